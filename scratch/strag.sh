set -e
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3), d['parity'] if 'parity' in d else '')"; }
for i in 1 2; do
for v in 0 8 12 16 24 32; do MI_TRACER_STRAGGLERS=$v run strag $v; done
done
WL="--streams 1"
for v in 0 16 32; do MI_TRACER_STRAGGLERS=$v run one-stream strag $v; done
WL="--workload C5-glass-1024-16spp"
for v in 0 16 32; do MI_TRACER_STRAGGLERS=$v run C5 strag $v; done
WL="--workload C2-cornell-512-64spp"
for v in 0 16 32; do MI_TRACER_STRAGGLERS=$v run C2 strag $v; done
