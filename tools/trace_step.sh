#!/bin/bash
# Diagnostic: average duration of the training step's kernels (rocprofv3 kernel trace of a short bench run) under an optional env setting.
#   bash tools/trace_step.sh <out_dir> [VAR=value ...]
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fixed-job --no-k-sweep > $out/log.txt 2>&1
python3 - $out "$@" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("== ", " ".join(sys.argv[2:]) or "default")
tot = 0
for r in rows:
    if int(r["Calls"]) >= 1000:
        tot += float(r["AverageNs"])
        print(f"   {r['Name'][:70]:70s} {int(r['Calls']):6d} x {float(r['AverageNs'])/1000:7.2f} us")
print(f"   sum of the per-step kernels {tot/1000:.1f} us")
PY
