#!/bin/bash
# round 4, first GPU collection: the new tests, the clock read, the ingest thread sweep, the default bench line
set -o pipefail
O=gpurun_out/r04_a; mkdir -p $O
python -m pytest tests/test_gpu_mimic.py tests/test_cli_surface.py -m gpu -x -q -k "skewed or cfg3 or one_pass_slots" > $O/pytest_new.txt 2>&1 || { tail -30 $O/pytest_new.txt; exit 1; }
tail -3 $O/pytest_new.txt
timeout -k 10 300 python tools/mfma_clock.py > $O/mfma_clock.txt 2> $O/mfma_clock.err || { tail -20 $O/mfma_clock.err; exit 1; }
cat $O/mfma_clock.txt
timeout -k 10 400 python tools/sweep_ingest_threads.py > $O/ingest_threads.txt 2> $O/ingest_threads.err || { tail -20 $O/ingest_threads.err; exit 1; }
cat $O/ingest_threads.txt
timeout -k 10 500 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err || { tail -30 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04_a/bench_default.json"))
print({k: j[k] for k in ("value", "ms_per_step", "job")}, j["stage_ms"], j.get("t_e2e"))
print(json.dumps(j.get("fixed_job_8_voters"), indent=1)[:1500])
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j["cpu_baseline"]["value_reference_threads"])
PY
