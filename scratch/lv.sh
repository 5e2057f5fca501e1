# usage: bash scratch/lv.sh "<ENV=VAL ...>" ...   — per-level timings of the C3 frame under each environment
for e in "$@"; do
  echo "== $e"
  env $e MI_TRACER_DEBUG_LEVELS=1 timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> /tmp/lv.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || exit 1
  grep "^batch" /tmp/lv.err | tail -7 | head -4
done
