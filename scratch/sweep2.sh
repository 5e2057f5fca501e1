set -e
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2; do
run base
for v in 5 7; do MI_TRACER_EXTEND_WG_TWO_STREAMS=$v run wg_two $v; done
for v in 5 7; do MI_TRACER_EXTEND_WG_PRIMARY=$v run wg_primary $v; done
for v in 16 24; do MI_TRACER_REFILL=$v run refill $v; done
for v in 12 20; do MI_TRACER_LEAF_MIN=$v run leaf_min $v; done
for v in 8 16; do MI_TRACER_REFILL_PRIMARY=$v run refill_primary $v; done
done
