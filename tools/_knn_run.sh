#!/bin/bash
# one GPU call: knn tests, then the 10^6 timing with a kernel trace
set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_knn.py -x -q -s > gpurun_out/knn_tests.log 2>&1 || { tail -40 gpurun_out/knn_tests.log; exit 1; }
tail -15 gpurun_out/knn_tests.log
timeout -k 10 400 python3 tools/bench_knn.py > gpurun_out/knn_bench.log 2>&1 || { tail -30 gpurun_out/knn_bench.log; exit 1; }
cat gpurun_out/knn_bench.log
