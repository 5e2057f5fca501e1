# usage: bash scratch/ab.sh <lib.so|base>[:ENV=VAL[,ENV=VAL]] ...  — C3 per-level timings with each library in place of
# libmi_tracer.so (GPU box copy only), two runs each
cp pathtracer_amd/lib/libmi_tracer.so /tmp/libmi_base.so
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  echo "=== $lib $envs"
  if [ "$lib" = base ]; then cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so; else cp $lib pathtracer_amd/lib/libmi_tracer.so; fi
  for rep in 1 2; do
  env $envs MI_TRACER_DEBUG_LEVELS=1 timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${WORKLOAD:+--workload $WORKLOAD} 2> /tmp/lv.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), end='  ')" || exit 1
  grep "^batch 0" /tmp/lv.err | head -14 | grep -v "level [456]" | awk '{printf "b%s L%s e %s s %s t %s | ", $2, $4, $9, $12, $6} END{print ""}'
  done
done
cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so
