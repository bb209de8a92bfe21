#!/bin/bash
# Diagnostic: per-kernel averages of the training step under a list of env settings, one rocprofv3 kernel trace each.
#   bash tools/ablate_step.sh <out_dir> "VAR=a" "VAR=b VAR2=c" ...     ("" = default)
out=$1; shift
mkdir -p $out
i=0
for setting in "$@"; do
    bash tools/trace_step.sh $out/run$i $setting | tee -a $out/summary.txt
    i=$((i+1))
done
