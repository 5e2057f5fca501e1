set -e
for cfg in "MI_TRACER_TAIL=5000000" "MI_TRACER_TAIL_WG_PER_CU=3" "MI_TRACER_TAIL_WG_PER_CU=2" "MI_TRACER_SHADE_MIN=16" "MI_TRACER_SHADE_MIN=32" "MI_TRACER_TAIL=800000" "MI_TRACER_TAIL=2500000"; do
  echo "== $cfg"
  env $cfg timeout -k 10 60 python scratch/dbg.py 2>&1 | tail -1
done
