set -e
for v in 0 32 512 8192; do
  echo "== PT_BVH_SWEEP=$v"
  PT_BVH_SWEEP=$v timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), 'sah', round(d['config']['bvh_sah_cost'],2), 'depth', d['config']['bvh_depth'], 'nodes/ray', round(d['roofline']['nodes_per_ray'],2), 'prims/ray', round(d['roofline']['prim_tests_per_ray'],2))"
done
