# usage: bash scratch/regs.sh [extra -D flags] — VGPRs, spills, LDS, occupancy of the persistent kernels as compiled now
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/regs.o pathtracer_amd/csrc/mi_tracer.hip 2>&1 | \
 grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//' | paste - - - - - - | grep -E "k_extend|k_tail|k_shade" | sed 's/  */ /g'
