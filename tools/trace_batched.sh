#!/bin/bash
# Diagnostic: per-launch averages of the batched-voter step (rocprofv3 kernel trace of tools/concurrent_voters.py).
#   bash tools/trace_batched.sh <out_dir> <voters> [VAR=value ...]
out=$1; shift
v=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
env IDELUCS_TUNABLEOP=1 "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/concurrent_voters.py --n 50000 --voters $v --reps 2 > $out/log.txt 2>&1
grep -a voter-epochs $out/log.txt | tail -3
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 500:
        print(f"   {r['Name'][:100]:100s} {int(r['Calls']):6d} x {float(r['AverageNs'])/1000:7.2f} us")
PY
rm -f $out/*/*kernel_trace.csv
