#!/bin/bash
# Latency and instruction-class counters of one bench configuration (GPU box, repo root).
# usage: tools/pmc_lat.sh <tag> [bench.py args...]  -> gpurun_out/pmc_lat_<tag>/{d,e,f,g}
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_lat_$tag
mkdir -p $out
ARGS="$*"
run() {
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $ARGS > $out/$name.log 2>&1
}
run d SQ_INST_LEVEL_VMEM SQ_ACCUM_PREV_HIRES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run e SQ_INST_LEVEL_LDS SQ_ACCUM_PREV_HIRES SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE
run f SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU
run g SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
find $out -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $out k_edge_bwd k_edge_fwd k_node_gather
