# usage: bash scratch/wl.sh "<ENV=VAL,...>" ...  — Mrays/s of the four workloads under each environment
for e in "$@"; do
  envs=$(echo $e | tr ',' ' ')
  for wl in C2-cornell-512-64spp C3-blob82k-1080p-16spp C4-blob1.3M-4k-256spp C5-glass-1024-16spp; do
    steps=10; [ $wl = C4-blob1.3M-4k-256spp ] && steps=3
    env $envs timeout -k 10 300 python bench.py --steps $steps --warmup 2 --no-cpu-baseline --workload $wl 2> /tmp/wl.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $wl Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || { tail -5 /tmp/wl.err; exit 1; }
  done
done
