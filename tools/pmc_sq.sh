#!/bin/bash
# SQ counter passes over one bench configuration (run on the GPU box from the repo root).
# usage: tools/pmc_sq.sh <tag> [bench.py args...]    -> gpurun_out/pmc_<tag>/{a,b,c}
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
run() {
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $ARGS > $out/$name.log 2>&1
}
ARGS="$*"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES
run c SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT
tail -n 2 $out/*.log
