#!/bin/bash
# HBM traffic of the dominant kernels from rocprofv3 PMC counters (run on the GPU box):
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md §HBM / PMC slots), same
# bench command as the timing run, plus the memory-side request counters by size
# (TCC_EA0_RDREQ{,_32B,_64B,_128B}, TCC_EA0_WRREQ{,_64B}) that say how many bytes each request moved -
# the cross-check of the guide's "double FETCH_SIZE for 16-byte-per-lane reads" correction.
# Writes profiles/<tag>_traffic.json.   Usage: tools/measure_traffic.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/traffic_$TAG
rm -rf $OUT
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > /dev/null 2>&1
}
ARGS="$*"
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass RDREQ TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass RDREQ2 TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass WRREQ TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
find $OUT -name "*kernel_trace.csv" -delete
python3 $R/tools/parse_traffic.py "$OUT" "$R/gpurun_out/${TAG}_traffic.json" "$@"
cp "$R/gpurun_out/${TAG}_traffic.json" "$R/profiles/${TAG}_traffic.json" 2>/dev/null || true
