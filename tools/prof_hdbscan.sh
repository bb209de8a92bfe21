#!/bin/bash
# rocprofv3 kernel statistics of the device HDBSCAN (tools/bench_hdbscan.py N, Gaussian blobs): the top kernels -> gpurun_out/
#   bash tools/prof_hdbscan.sh [N]
set -e
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_hdb
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hdb -- python3 $root/tools/bench_hdbscan.py ${1:-1000000} > $out/prof_hdbscan.log 2>&1 || { tail -20 $out/prof_hdbscan.log; exit 1; }
grep "n = " $out/prof_hdbscan.log
head -12 /tmp/prof_hdb/*/*kernel_stats.csv | cut -c1-260 > $out/prof_hdbscan_kernel_stats.csv
cat $out/prof_hdbscan_kernel_stats.csv | cut -c1-200
