set -e
timeout -k 10 600 python scratch/sah_eq.py big 2>&1 | grep -v IDENTICAL || true
for wl in C3 C4 C5 C2; do
timeout -k 10 200 python scratch/treecmp.py $wl host 2>&1 | tail -1
timeout -k 10 200 python scratch/treecmp.py $wl device 2>&1 | tail -1
MI_TRACER_BUILDER=ploc timeout -k 10 200 python scratch/treecmp.py $wl device 2>&1 | tail -1
done
