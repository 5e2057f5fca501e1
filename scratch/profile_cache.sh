# usage: bash scratch/profile_cache.sh <tag>   (GPU box, repo root): L2 / L1 hit rates and VALU lane activity of one C3 frame
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc_tcc_$TAG -o run --output-format csv -- python $R/scratch/one.py > /dev/null 2> $R/gpurun_out/pmc_tcc_${TAG}.log
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc_tcp_$TAG -o run --output-format csv -- python $R/scratch/one.py > /dev/null 2> $R/gpurun_out/pmc_tcp_${TAG}.log
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT -d $R/gpurun_out/pmc_lane_$TAG -o run --output-format csv -- python $R/scratch/one.py > /dev/null 2> $R/gpurun_out/pmc_lane_${TAG}.log
cd $R
for k in tcc tcp lane; do python scratch/pmc_parse2.py gpurun_out/pmc_${k}_$TAG > gpurun_out/${k}_${TAG}.txt; done
head -3 gpurun_out/tcc_${TAG}.txt
