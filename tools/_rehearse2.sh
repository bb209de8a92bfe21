#!/bin/bash
# diagnostic: two ranks of bench.py on this box's one GPU over gloo, small input; prints ms per step and the stage times
export IDELUCS_BENCH_BACKEND=gloo IDELUCS_BENCH_DEVICES=1
timeout -k 10 ${2:-100} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29615 bench.py --gpus 2 --n-sequences ${1:-5000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
j = json.loads(l[-1]); print(round(j['ms_per_step'], 1), {k: round(x, 1) for k, x in j['stage_ms'].items()})"
