#!/bin/bash
# sweep of tail knobs with the both-phase tail (N=1 frame and 1/8 shard)
for sm in 24 8 1; do for tt in 1500000 3000000 5000000; do
  echo "shade_min $sm tail $tt"
  MI_TAIL_BOTH=1 MI_TRACER_SHADE_MIN=$sm MI_TRACER_TAIL=$tt timeout -k 10 100 python scratch/dbg.py 2>&1 | grep -E "total" || exit 1
  MI_TAIL_BOTH=1 MI_TRACER_SHADE_MIN=$sm MI_TRACER_TAIL=$tt timeout -k 10 100 python scratch/shard8.py 8 2>&1 | grep -E "total" || exit 1
done; done
