# usage: bash scratch/ab_env.sh "<ENV=VAL>" [rounds]   — bench.py alternately without and with the variable, same box, same call
V=$1; N=${2:-4}
for i in $(seq $N); do
  for e in "X_=0" "$V"; do
    env $e timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$e',round(d['value']),round(d['ms_per_step'],3))" || exit 1
  done
done
