mkdir -p gpurun_out/mid
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_encoder.py -q -m gpu -x 2>&1 | tail -2 > gpurun_out/mid/test.txt
for rep in 1 2; do for v in 0 1; do
IDELUCS_MID_FUSED=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/mid/bench_mid${v}_r${rep}.json
done; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
export IDELUCS_MID_FUSED=$v
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mid$v -o mid -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/mid/prof$v.log 2>&1
find /tmp/prof_mid$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/mid/kernel_stats$v.csv
done
