#!/bin/bash
# Diagnostic: bench.py (no CPU baseline, no e2e) under values of ONE environment variable:  bash tools/sweep_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['value']), 'seq/s  epoch', round(d['stage_ms']['epoch'],2), 'ms  vectorise', round(d['stage_ms']['vectorise'],3), 'ms')"
done
