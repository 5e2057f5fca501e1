for t in 700000 400000 200000 100000 50000; do
  echo "== TAIL=$t"; MI_TRACER_TAIL=$t timeout -k 10 120 python scratch/shards.py C3-blob82k-1080p-16spp 1,4,8 || exit 1
done
