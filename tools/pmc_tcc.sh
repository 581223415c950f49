#!/bin/bash
# Memory-side (fabric) request counters by size for one bench configuration (run on the GPU box from
# the repo root): bytes = 32 B x RDREQ_32B + 64 B x (RDREQ - RDREQ_32B), likewise for writes.
# This resolves what FETCH_SIZE alone cannot on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE tallies every
# request at 64 B): a 16-byte-per-lane streaming read issues 128-B requests, a scattered 16-B read
# does not. usage: tools/pmc_tcc.sh <tag> [bench.py args...]  -> gpurun_out/tcc_<tag>/
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/tcc_$tag
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_BUBBLE[A-Z0-9_]*" $out/counters.txt | sort -u > $out/tcc_names.txt
run() {
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $ARGS > $out/$name.log 2>&1
}
ARGS="$*"
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run rd2 TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run hit TCC_HIT_sum TCC_MISS_sum
find $out -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $out k_node_gather k_edge_bwd k_edge_fwd > $out/summary.txt 2>&1
cat $out/summary.txt
