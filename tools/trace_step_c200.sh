#!/bin/bash
# Diagnostic: per-kernel averages of the training step at n_clusters = 200 (the cfg5 / --n_clusters 0 shape), from a rocprofv3 kernel trace of
# tools/concurrent_voters.py run with one voter.
#   bash tools/trace_step_c200.sh <out_dir> [VAR=value ...]
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/concurrent_voters.py --n 60000 --voters 1 --reps 2 --n-clusters 200 > $out/log.txt 2>&1
grep -a voter-epochs $out/log.txt
python3 - $out "$@" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("== ", " ".join(sys.argv[2:]) or "default")
tot = 0
for r in rows:
    if int(r["Calls"]) >= 1000:
        tot += float(r["TotalDurationNs"])
        print(f"   {r['Name'][:90]:90s} {int(r['Calls']):6d} x {float(r['AverageNs'])/1000:7.2f} us")
print(f"   total of these kernels {tot/1e6:.1f} ms")
PY
