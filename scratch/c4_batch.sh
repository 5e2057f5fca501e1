for bp in 0 134217728 268435456; do
  timeout -k 10 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --workload C4-blob1.3M-4k-256spp --batch-paths $bp 2> /tmp/c4.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch_paths $bp Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || { tail -5 /tmp/c4.err; exit 1; }
done
