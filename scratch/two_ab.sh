set -e
L=$PWD/pathtracer_amd/libtail2
MI_TRACER_LIB_DIR=$L timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_levels.py tests/test_gpu_full_size.py tests/test_gpu_edge_cases.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run base; MI_TRACER_LIB_DIR=$L run tail2; done
WL="--streams 1"; run base one-stream; MI_TRACER_LIB_DIR=$L run tail2 one-stream
WL="--workload C5-glass-1024-16spp"; run base C5; MI_TRACER_LIB_DIR=$L run tail2 C5
WL="--workload C2-cornell-512-64spp"; run base C2; MI_TRACER_LIB_DIR=$L run tail2 C2
