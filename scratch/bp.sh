for bp in 0 11100000 8300000 5600000 4200000; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch-paths $bp 2> /tmp/wl.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch_paths $bp Mrays/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))" || { tail -5 /tmp/wl.err; exit 1; }
done
