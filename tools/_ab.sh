mkdir -p gpurun_out/ab
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | tail -2 > gpurun_out/ab/test.txt
for rep in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab/bench_r${rep}.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o ab -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ab/prof.log 2>&1
find /tmp/prof_ab -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/ab/kernel_stats.csv
