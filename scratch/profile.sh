# usage: bash scratch/profile.sh <tag>   (run on the GPU box from the repo root)
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_fetch_${TAG}.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$TAG -o run --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_write_${TAG}.log
cd $R
ls gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
cat gpurun_out/prof_${TAG}_bench.json | cut -c1-200
# SQ counters per dispatch of one C3 frame (8 SQ counters = one pass)
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_sq_$TAG -o run --output-format csv -- python $R/scratch/one.py > /dev/null 2> $R/gpurun_out/pmc_sq_${TAG}.log
cd $R
python scratch/pmc_parse2.py gpurun_out/pmc_sq_$TAG > gpurun_out/sq_${TAG}.txt
