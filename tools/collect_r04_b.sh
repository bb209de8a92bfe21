#!/bin/bash
# round 4, second collection: the own layer-1 forward in the step
set -o pipefail
O=gpurun_out/r04_b; mkdir -p $O
python -m pytest tests/test_gpu_encoder.py -m gpu -x -q > $O/pytest_encoder.txt 2>&1 || { tail -40 $O/pytest_encoder.txt; exit 1; }
tail -3 $O/pytest_encoder.txt
bash tools/trace_step.sh $O/trace_l1 > $O/step_kernels_l1.txt 2>&1; cat $O/step_kernels_l1.txt
bash tools/trace_step.sh $O/trace_lib IDELUCS_L1_FUSED=0 > $O/step_kernels_lib.txt 2>&1; cat $O/step_kernels_lib.txt
for v in 1 0; do IDELUCS_L1_FUSED=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fixed-job > $O/bench_l1_$v.json 2> $O/bench_l1_$v.err || { tail -30 $O/bench_l1_$v.err; exit 1; }
python - $O/bench_l1_$v.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print(sys.argv[1], j["value"], j["ms_per_step"], j["stage_ms"])
PY
done
