# usage: tools/_ab.sh VAR v1 v2 ...   -- interleaved A/B of bench.py under an environment variable, two rounds
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$var=$v', round(j['value']), round(j['stage_ms']['epoch'],2))"
done; done
