set -e
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3), d.get('parity',{}).get('radiance_bit_identical'))"; }
for i in 1 2; do
run base
for v in 2_8 2_16 3_8; do MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/librep_$v run rep $v; done
done
WL="--streams 1"
run base one-stream
for v in 2_8 2_16 3_8; do MI_TRACER_LIB_DIR=$PWD/pathtracer_amd/librep_$v run one-stream rep $v; done
