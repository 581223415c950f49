#!/bin/bash
# The bench lines, rocprofv3 kernel summaries and HBM-traffic counters that profiles/ keeps
# (run on the GPU box from the repo root). usage: tools/final_pass.sh <tag>  -> gpurun_out/final_<tag>/
tag=$1
out=gpurun_out/final_$tag
mkdir -p $out
export TMPDIR=/tmp
line() { grep '^{' | tail -1; }          # keep the JSON line only ([Gloo] / library chatter goes to stdout too)
python3 bench.py --steps 20 --warmup 5 2> $out/bench_cfg2.err | line > $out/bench_cfg2.json
python3 bench.py --config cfg3 2>/dev/null | line > $out/bench_cfg3.json        # (with cpu_baseline since round 4)
python3 bench.py --infer --no-cpu-baseline 2>/dev/null | line > $out/bench_cfg2_infer.json
python3 bench.py --config cfg5 --sweep 12800 2>/dev/null | line > $out/bench_cfg5.json
python3 bench.py --skip-dead-coords --no-cpu-baseline 2>/dev/null | line > $out/bench_cfg2_skip.json
PVS_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | line > $out/bench_cfg2_gpus2_gloo_shared_gpu.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_cfg2.json 2> $out/prof_cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cfg3 > $out/prof_cfg3.json 2> $out/prof_cfg3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg5 -- python3 bench.py --config cfg5 --steps 50 --warmup 5 --graph 0 --no-cpu-baseline > $out/prof_cfg5.json 2> $out/prof_cfg5.err
# one step's launches in order (round 6: kept for cfg2 and cfg3 too), then keep only the summaries (the traces are large)
python3 tools/step_timeline.py $out/prof_cfg2 > $out/step_timeline_cfg2.txt 2>&1
python3 tools/step_timeline.py $out/prof_cfg3 > $out/step_timeline_cfg3.txt 2>&1
find $out -name '*kernel_trace.csv' -delete
tools/measure_traffic.sh ${tag}_cfg2 > $out/traffic_cfg2.log 2>&1
tools/measure_traffic.sh ${tag}_cfg5 --config cfg5 --steps 5 --graph 0 > $out/traffic_cfg5.log 2>&1
tools/measure_traffic.sh ${tag}_cfg3 --config cfg3 > $out/traffic_cfg3.log 2>&1
tools/pmc_sq.sh ${tag} > $out/pmc_sq.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${tag} k_edge_bwd k_edge_fwd k_node_gather > $out/pmc_sq_summary.txt 2>&1
tools/pmc_sq.sh ${tag}_cfg3 --config cfg3 > $out/pmc_sq_cfg3.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_cfg3 k_edge_bwd k_edge_fwd k_node_gather > $out/pmc_sq_cfg3_summary.txt 2>&1
find gpurun_out/pmc_${tag} gpurun_out/pmc_${tag}_cfg3 gpurun_out/traffic_${tag}_cfg2 gpurun_out/traffic_${tag}_cfg5 gpurun_out/traffic_${tag}_cfg3 -name '*.csv' -size +1M -delete
for f in $out/bench_*.json $out/prof_cfg*.json; do echo "$f: $(cut -c1-260 $f | grep -o '"value": [0-9.]*')"; done
# round 3 additions: soak (reproducibility of every kernel family), latency / instruction-class counters, variants
python3 tools/soak.py --repeats 300 --states clean --out $out/soak_alone.json > $out/soak_alone.log 2>&1
python3 tools/soak.py --repeats 300 --states clean --load --out $out/soak_load.json > $out/soak_load.log 2>&1
python3 tools/soak.py --repeats 40 --states nan,garbage --out $out/soak_poison.json > $out/soak_poison.log 2>&1
tools/pmc_lat.sh ${tag} > $out/pmc_lat.txt 2>&1
find gpurun_out/pmc_lat_${tag} -name '*.csv' -size +1M -delete
tools/variants_r3b.sh $out/variants.txt > /dev/null 2>&1
python3 bench.py --config cfg5 --sweep 100000 --no-cpu-baseline 2>/dev/null | line > $out/bench_cfg5_sweep100000.json
# round 4 additions: sustained clocks, strong-scaling legs, the 1-rank RCCL all-reduce, strict-parity margins, dynamic range
tools/sustained.sh ${tag} > $out/sustained_summary.txt 2>&1
tools/strong_scaling_legs.sh > $out/strong_scaling_one_gpu_legs.txt 2>&1
python3 tools/allreduce_microbench.py 2>/dev/null | line > $out/allreduce_one_rank_rccl.json
python3 tools/dynamic_range_probe.py 2>/dev/null | grep -v amdgpu.ids > $out/dynamic_range.txt
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py -q -m gpu > $out/parity.log 2>&1; cp gpurun_out/parity_margins.txt $out/parity_margins.txt
# round 5 additions: the launch-bound shape (the reference's CLI defaults), eager and captured, with one step's timeline
python3 bench.py --config real4A --steps 200 --warmup 30 --graph 0 --no-cpu-baseline 2>/dev/null | line > $out/real_shape_eager.json
python3 bench.py --config real4A --steps 200 --warmup 30 --graph 1 --no-cpu-baseline 2>/dev/null | line > $out/real_shape_graph.json
rocprofv3 --kernel-trace --output-format csv -d $out/trace_real4A -- python3 bench.py --config real4A --steps 3 --warmup 2 --graph 0 --no-cpu-baseline > /dev/null 2>&1
python3 tools/step_timeline.py $out/trace_real4A > $out/step_timeline_real4A.txt 2>&1
find $out/trace_real4A -name '*.csv' -size +1M -delete
python3 -m pytest tests/test_gpu_lazy_scales.py tests/test_gpu_properties.py -q -m gpu -k "lazy or tile_magnitudes or bias_sums or dynamic_range or binades" > $out/lazy_tests.log 2>&1
tools/micro/valu_issue_bench.bin sigmoid > $out/micro_sigmoid.txt 2>&1

# the H = 32 backward phase by phase (needs the -DPVS_TILE_TRACE build: tools/variant_obj.sh trace edge_bwd_f16.hip "-DPVS_TILE_TRACE"), LDS-DMA addressing
[ -f pointvs_amd/libpvs_egnn_trace.so ] && PVS_EGNN_LIB=pointvs_amd/libpvs_egnn_trace.so python3 tools/tile_trace.py 2>&1 | grep -v amdgpu.ids > $out/tile_trace.txt
[ -x tools/micro/glds_offset_test.bin ] && tools/micro/glds_offset_test.bin > $out/micro_glds_offset.txt 2>&1

# round 6 additions: which host operation launches the torch-side kernels of a step; the gloo two-rank line under an RCCL protocol flag (recorded only)
python3 tools/launch_origins.py cfg2 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $out/launch_origins_cfg2.txt
python3 bench.py --gpus 1 --force-dist --rccl-proto LL --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line > $out/bench_cfg2_one_rank_rccl_ll.json
