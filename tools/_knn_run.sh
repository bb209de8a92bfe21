#!/bin/bash
# one GPU call: knn tests, a 300 000-sequence cfg5 run that leaves its latent, diagnostics on that latent
set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_knn.py -x -q -s > gpurun_out/knn_tests.log 2>&1 || { tail -40 gpurun_out/knn_tests.log; exit 1; }
tail -12 gpurun_out/knn_tests.log
export IDELUCS_TIMING=1
timeout -k 10 400 python3 tools/run_cfg5_cli.py --n ${1:-300000} --save-latent /dev/shm/latent.npy > gpurun_out/cfg5_small.log 2>&1 || { tail -30 gpurun_out/cfg5_small.log; exit 1; }
grep -E "HDBSCAN on|stages|wall" gpurun_out/cfg5_small.log
timeout -k 10 500 python3 tools/diag_hdbscan.py /dev/shm/latent.npy --compare > gpurun_out/diag_hdbscan.log 2>&1 || { tail -30 gpurun_out/diag_hdbscan.log; rm -f /dev/shm/latent.npy; exit 1; }
rm -f /dev/shm/latent.npy
cat gpurun_out/diag_hdbscan.log
