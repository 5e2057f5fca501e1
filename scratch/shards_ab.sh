# usage: bash scratch/shards_ab.sh <lib.so|base> ... — C3 shard times (N = 1, 8) and C5 with each library in place of libmi_tracer.so
cp pathtracer_amd/lib/libmi_tracer.so /tmp/libmi_base.so
for lib in "$@"; do
  echo "=== $lib"
  if [ "$lib" = base ]; then cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so; else cp $lib pathtracer_amd/lib/libmi_tracer.so; fi
  for rep in 1 2; do timeout -k 10 200 python scratch/shards.py C3-blob82k-1080p-16spp 1,8 || exit 1; done
  timeout -k 10 200 python scratch/shards.py C5-glass-1024-16spp 1 || exit 1
done
cp /tmp/libmi_base.so pathtracer_amd/lib/libmi_tracer.so
