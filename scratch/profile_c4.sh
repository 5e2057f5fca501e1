# usage: bash scratch/profile_c4.sh   (GPU box, repo root): SQ + L2 counters of one C4 frame
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_c4_sq -o run --output-format csv -- python $R/scratch/c4.py > /dev/null 2> $R/gpurun_out/pmc_c4_sq.log
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_c4_tcc -o run --output-format csv -- python $R/scratch/c4.py > /dev/null 2> $R/gpurun_out/pmc_c4_tcc.log
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc_c4_tcp -o run --output-format csv -- python $R/scratch/c4.py > /dev/null 2> $R/gpurun_out/pmc_c4_tcp.log
cd $R
for k in sq tcc tcp; do python scratch/pmc_parse2.py gpurun_out/pmc_c4_$k > gpurun_out/c4_$k.txt; done
