# usage: bash scratch/ab2.sh <workload> <lib.so|base> ... — Mrays/s without per-launch events, three runs each, libraries interleaved
wl=$1; shift
cp pathtracer_amd/lib/libmi_tracer.so /tmp/libmi_base.so
for rep in 1 2 3; do
for lib in "$@"; do
  if [ "$lib" = base ]; then cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so; else cp $lib pathtracer_amd/lib/libmi_tracer.so; fi
  steps=15; [ $wl = C4-blob1.3M-4k-256spp ] && steps=3
  timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --workload $wl 2> /tmp/wl.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $wl Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || { tail -5 /tmp/wl.err; cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so; exit 1; }
done; done
cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so
