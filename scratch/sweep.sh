# usage: bash scratch/sweep.sh <workload> "<ENV=VAL,...>" ... — Mrays/s and ms of one workload under each environment, two runs each
wl=$1; shift
for e in "$@"; do
  envs=$(echo $e | tr ',' ' ')
  for rep in 1 2; do
    steps=15; [ $wl = C4-blob1.3M-4k-256spp ] && steps=3
    env $envs timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --workload $wl 2> /tmp/wl.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $wl Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || { tail -5 /tmp/wl.err; exit 1; }
  done
done
