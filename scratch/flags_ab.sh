set -e
L=$PWD/pathtracer_amd/libf_noslp
run() { python bench.py --steps $ST --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3))"; }
ST=20
WL="--streams 1"; run base one-stream; MI_TRACER_LIB_DIR=$L run noslp one-stream
WL="--workload C5-glass-1024-16spp"; run base C5; MI_TRACER_LIB_DIR=$L run noslp C5; run base C5; MI_TRACER_LIB_DIR=$L run noslp C5
WL="--workload C2-cornell-512-64spp"; run base C2; MI_TRACER_LIB_DIR=$L run noslp C2; run base C2; MI_TRACER_LIB_DIR=$L run noslp C2
ST=3; WL="--workload C4-blob1.3M-4k-256spp"; run base C4; MI_TRACER_LIB_DIR=$L run noslp C4
