set -e
for cfg in "MI_TRACER_LEAF_MIN=8" "MI_TRACER_LEAF_MIN=16" "MI_TRACER_LEAF_MIN=24" "MI_TRACER_LEAF_MIN=32" "MI_TRACER_REFILL=12" "MI_TRACER_REFILL=28" "MI_TRACER_EXTEND_WG_PER_CU=5" "MI_TRACER_EXTEND_WG_PER_CU=7"; do
  echo "== $cfg"
  env $cfg timeout -k 10 60 python scratch/dbg.py 2>&1 | tail -1
done
