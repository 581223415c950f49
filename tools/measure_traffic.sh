#!/bin/bash
# HBM traffic of the dominant kernels from rocprofv3 PMC counters (run on the GPU box):
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md §HBM / PMC slots), same
# bench command as the timing run. Writes profiles/<tag>_traffic.json.
# Usage: tools/measure_traffic.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/traffic_$TAG
rm -rf $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
done
python3 $R/tools/parse_traffic.py "$OUT" "$R/gpurun_out/${TAG}_traffic.json" "$@"
cp "$R/gpurun_out/${TAG}_traffic.json" "$R/profiles/${TAG}_traffic.json" 2>/dev/null || true
