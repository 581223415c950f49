#!/bin/bash
# Refresh the bench lines and kernel summaries of profiles/<tag>_* after a change of bench.py alone (kernels unchanged:
# the PMC, traffic and soak files of tools/final_pass.sh stay valid).   usage (GPU box): tools/refresh_bench_lines.sh <tag>
tag=$1; out=gpurun_out/final_$tag; mkdir -p $out; export TMPDIR=/tmp
line() { grep '^{' | tail -1; }
python3 bench.py --steps 20 --warmup 5 2>/dev/null | line > $out/bench_cfg2.json
python3 bench.py --config cfg3 2>/dev/null | line > $out/bench_cfg3.json
python3 bench.py --infer --no-cpu-baseline 2>/dev/null | line > $out/bench_cfg2_infer.json
python3 bench.py --skip-dead-coords --no-cpu-baseline 2>/dev/null | line > $out/bench_cfg2_skip.json
PVS_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | line > $out/bench_cfg2_gpus2_gloo_shared_gpu.json
rm -rf $out/prof_cfg2 $out/prof_cfg3
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_cfg2.json 2> $out/prof_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cfg3 > $out/prof_cfg3.json 2> $out/prof_cfg3.err
find $out -name '*kernel_trace.csv' -delete
tools/sustained.sh ${tag} > $out/sustained_summary.txt 2>&1
tools/strong_scaling_legs.sh > $out/strong_scaling_one_gpu_legs.txt 2>&1
for f in $out/bench_*.json $out/prof_cfg[23].json; do echo "$f: $(cut -c1-260 $f | grep -o '"value": [0-9.]*')"; done
