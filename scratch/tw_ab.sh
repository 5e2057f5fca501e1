set -e
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run base; for v in 4 6; do MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/libtw$v run tailwg $v; done; done
WL="--streams 1"; run base one-stream; for v in 4 6; do MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/libtw$v run one-stream tailwg $v; done
WL="--workload C5-glass-1024-16spp"; run base C5; for v in 4 6; do MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/libtw$v run C5 tailwg $v; done
WL="--workload C2-cornell-512-64spp"; run base C2; for v in 4 6; do MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/libtw$v run C2 tailwg $v; done
