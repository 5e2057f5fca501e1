set -e
for v in 4 5 8 10 15 20; do
  echo "== MI_TRACER_SHADE_WG_PER_CU=$v"
  MI_TRACER_SHADE_WG_PER_CU=$v timeout -k 10 60 python scratch/dbg.py 2>&1 | tail -1
done
