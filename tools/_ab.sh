mkdir -p gpurun_out/kb
python -m pytest tests -q -m gpu -x 2>&1 | tail -2 > gpurun_out/kb/test.txt
python tools/bench_kernels.py > gpurun_out/kb/k.txt 2>&1
for rep in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/kb/bench_r${rep}.json
done
