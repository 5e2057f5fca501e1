set -e
MI_TRACER_TAIL_REQUEUE=1 timeout -k 10 200 python -u -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -1
for cfg in "MI_TRACER_TAIL_REQUEUE=0" "MI_TRACER_TAIL_REQUEUE=1" "MI_TRACER_TAIL_REQUEUE=1 MI_TRACER_TAIL=5000000" "MI_TRACER_TAIL_REQUEUE=1 MI_TRACER_SHADE_MIN=16" "MI_TRACER_TAIL_REQUEUE=1 MI_TRACER_SHADE_MIN=40"; do
  echo "== $cfg"
  env $cfg timeout -k 10 60 python scratch/dbg.py 2>&1 | tail -1
  env $cfg timeout -k 10 60 python scratch/shard.py 2>&1 | grep "^N=8"
done
