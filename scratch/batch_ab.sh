set -e
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $* 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2; do
run
run --batch-paths 8400000
run --batch-paths 11100000
run --batch-paths 5600000
run --batch-paths 4200000
done
