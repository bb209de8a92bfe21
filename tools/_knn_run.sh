#!/bin/bash
# one GPU call: a cfg5 run that leaves its latent, then diagnostics on that latent
set -e
mkdir -p gpurun_out
export IDELUCS_TIMING=1
timeout -k 10 400 python3 tools/run_cfg5_cli.py --n ${1:-300000} --save-latent /dev/shm/latent.npy > gpurun_out/cfg5_small.log 2>&1 || { tail -30 gpurun_out/cfg5_small.log; exit 1; }
grep -E "HDBSCAN on|stages|wall" gpurun_out/cfg5_small.log
timeout -k 10 500 python3 ${2:-tools/diag_hdbscan.py} /dev/shm/latent.npy --compare > gpurun_out/diag_hdbscan.log 2>&1 || { tail -30 gpurun_out/diag_hdbscan.log; rm -f /dev/shm/latent.npy; exit 1; }
rm -f /dev/shm/latent.npy
cat gpurun_out/diag_hdbscan.log
