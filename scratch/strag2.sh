set -e
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/libhead run HEAD
MI_TRACER_STRAGGLERS=0 run new strag 0
MI_TRACER_STRAGGLERS=16 run new strag 16
done
WL="--streams 1"
MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/libhead run HEAD one stream
MI_TRACER_STRAGGLERS=0 run new strag 0 one stream
