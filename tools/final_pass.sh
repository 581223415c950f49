#!/bin/bash
# The bench lines, rocprofv3 kernel summaries and HBM-traffic counters that profiles/ keeps
# (run on the GPU box from the repo root). usage: tools/final_pass.sh <tag>  -> gpurun_out/final_<tag>/
tag=$1
out=gpurun_out/final_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err
python3 bench.py --config cfg3 --no-cpu-baseline > $out/bench_cfg3.json 2>/dev/null
python3 bench.py --infer --no-cpu-baseline > $out/bench_cfg2_infer.json 2>/dev/null
python3 bench.py --config cfg5 --steps 100 --warmup 10 > $out/bench_cfg5.json 2>/dev/null
python3 bench.py --skip-dead-coords --no-cpu-baseline > $out/bench_cfg2_skip.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_cfg2.json 2> $out/prof_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cfg3 > $out/prof_cfg3.json 2> $out/prof_cfg3.err
# keep only the summaries (the traces are large)
find $out -name '*kernel_trace.csv' -delete
tools/measure_traffic.sh ${tag}_cfg2 > $out/traffic.log 2>&1
for f in $out/bench_*.json $out/prof_cfg*.json; do echo "$f: $(cut -c1-260 $f | grep -o '"value": [0-9.]*')"; done
