# usage (on the GPU box, from the repo root):  bash profiles/tools/collect.sh <tag> [workload] [variant]
# (variant, e.g. wide8 with MI_TRACER_LIB_DIR=.../lib8 in the environment: profiles/roofline_pmc.json is left alone)
# Raw rocprofv3 output goes to gpurun_out/<tag>/; profiles/tools/pmc_to_json.py turns it into the JSON bench.py reads.
# Passes (each its own run, counters never combined with a trace):
#   stats : --kernel-trace --stats of `bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline`: every batch on one
#           stream, each launch alone on the chip (per-level durations for the JSON; k_extend's average = roofline.avg_launch_ms)
#   stats2: the same of the default command `bench.py --steps 10 --warmup 3 --no-cpu-baseline` (two streams; launches overlap)
#   fetch / write : HBM-side bytes (FETCH_SIZE, WRITE_SIZE)          of ONE frame (profiles/tools/one_frame.py)
#   sq    : VALU issue and lane counters, wave cycles                 "
#   tcc / tcp : L2 and vector-L1 requests                             "
#   grbm  : GRBM_GUI_ACTIVE (effective clock)                          "
set -e
TAG=$1
WL=${2:-C3-blob82k-1080p-16spp}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats_passes() {
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python $R/bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --workload $WL > $OUT/bench_profiled_one_stream.json 2> $OUT/stats.log
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats2 -o run --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $WL > $OUT/bench_profiled.json 2> $OUT/stats2.log
}
stats_passes
pass() { # name, counters...
    n=$1; shift
    timeout -s KILL 180 rocprofv3 --pmc "$@" -d $OUT/$n -o run --output-format csv -- python $R/profiles/tools/one_frame.py $WL > $OUT/$n.out 2> $OUT/$n.log
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY || true
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass td TD_TD_BUSY_sum TA_TA_BUSY_sum || true
pass grbm GRBM_GUI_ACTIVE || true
cd $R
python profiles/tools/pmc_to_json.py $OUT $TAG $WL $3
# The two bench lines above were printed while profiles/roofline_pmc.json still described the previous sources (their
# roofline is withheld as stale after a kernel edit): print them again, now against the pass just converted, and convert
# once more so that the per-level durations in the JSON are those of the kernel_stats files kept beside it.
if [ -z "$3" ]; then
    cd /tmp
    stats_passes
    cd $R
    python profiles/tools/pmc_to_json.py $OUT $TAG $WL
fi
