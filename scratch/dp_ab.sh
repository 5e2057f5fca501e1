set -e
run() { python bench.py --steps $ST --warmup 3 --no-cpu-baseline --bvh host --workload $WL 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$WL $*', round(d['ms_per_step'],3))"; }
for WL in C3-blob82k-1080p-16spp C2-cornell-512-64spp C5-glass-1024-16spp C4-blob1.3M-4k-256spp; do
ST=20; [ $WL = C4-blob1.3M-4k-256spp ] && ST=3
for i in 1 2 3; do run even; MI_TRACER_COLLAPSE=dp run dp; done
done
