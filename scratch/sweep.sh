set -e
for cfg in "MI_TRACER_EXTEND_WG_PER_CU=6" "MI_TRACER_EXTEND_WG_PER_CU=7" "MI_TRACER_EXTEND_WG_PER_CU=8" "MI_TRACER_EXTEND_WG_PER_CU=5" "MI_TRACER_REFILL=12" "MI_TRACER_REFILL=28"; do
  echo "== $cfg"
  env $cfg MI_TRACER_DEBUG_LEVELS=1 timeout -k 10 60 python scratch/dbg.py 2>&1 | tail -1
done
