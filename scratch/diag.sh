# usage: bash scratch/diag.sh <tag>  — diagnostic PMC passes of one C3 frame (vector memory / LDS pipeline pressure)
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" -d $OUT/$n -o run --output-format csv -- python $R/profiles/tools/one_frame.py > $OUT/$n.out 2> $OUT/$n.log || echo "pass $n failed"; }
pass d1 SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass d2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES
pass d3 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
pass d4 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass d5 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_BRANCH
cd $R
for n in d1 d2 d3 d4 d5; do echo "== $n"; python scratch/pmc_parse2.py $OUT/$n; done
