#!/bin/bash
# Copy the summaries of tools/final_pass.sh <tag> from gpurun_out/ into profiles/ (what the judge reads).
# usage: tools/collect_profiles.sh <tag>
tag=$1; src=gpurun_out/final_$tag; dst=profiles
cp $src/bench_cfg2.json $dst/${tag}_bench_cfg2.json
cp $src/bench_cfg3.json $dst/${tag}_bench_cfg3.json
cp $src/bench_cfg2_infer.json $dst/${tag}_bench_cfg2_infer.json
cp $src/bench_cfg5.json $dst/${tag}_bench_cfg5_sweep12800.json
cp $src/bench_cfg5_sweep100000.json $dst/${tag}_bench_cfg5_sweep100000.json 2>/dev/null
cp $src/bench_cfg2_skip.json $dst/${tag}_bench_cfg2_skip_dead_coords.json
cp $src/bench_cfg2_gpus2_gloo_shared_gpu.json $dst/${tag}_bench_cfg2_gpus2_gloo_shared_gpu.json
for c in cfg2 cfg3 cfg5; do
  cp $src/prof_$c.json $dst/${tag}_profiled_bench_$c.json
  f=$(ls -t $src/prof_$c/*/*kernel_stats.csv | head -1); cp $f $dst/${tag}_bench_${c}_kernel_stats.csv
done
cp $src/pmc_sq_summary.txt $dst/${tag}_cfg2_pmc_sq.txt
cp $src/pmc_sq_cfg3_summary.txt $dst/${tag}_cfg3_pmc_sq.txt
cp $src/pmc_lat.txt $dst/${tag}_cfg2_pmc_instruction_classes.txt
for c in cfg2 cfg3 cfg5; do cp gpurun_out/${tag}_${c}_traffic.json $dst/${tag}_${c}_traffic.json 2>/dev/null; done
cp $src/soak_alone.json $dst/${tag}_soak_alone.json
cp $src/soak_load.json $dst/${tag}_soak_concurrent_gpu_load.json
cp $src/soak_poison.json $dst/${tag}_soak_poisoned_allocator.json
cp $src/variants.txt $dst/${tag}_variants_backward_timings.txt
cp $src/sustained_summary.txt $dst/${tag}_sustained_summary.txt; for c in cfg2 cfg3; do grep '^{' gpurun_out/${tag}_sustained_$c.json | tail -1 > $dst/${tag}_sustained_$c.json; done
cp $src/strong_scaling_one_gpu_legs.txt $dst/${tag}_strong_scaling_one_gpu_legs.txt
cp $src/allreduce_one_rank_rccl.json $dst/${tag}_allreduce_one_rank_rccl.json
cp $src/dynamic_range.txt $dst/${tag}_dynamic_range.txt
cp $src/parity_margins.txt $dst/${tag}_parity_margins.txt
ls $dst | grep "^${tag}_" | wc -l
python3 tools/pmc_header.py $dst/${tag}_cfg2_pmc_sq.txt cfg2 > /dev/null; python3 tools/pmc_header.py $dst/${tag}_cfg3_pmc_sq.txt cfg3 > /dev/null
cp $src/real_shape_eager.json $dst/${tag}_real_shape_eager.json; cp $src/real_shape_graph.json $dst/${tag}_real_shape_graph.json; cp $src/step_timeline_real4A.txt $dst/${tag}_step_timeline_real4A.txt 2>/dev/null
# the commit the counters were taken at (bench.py: roofline.limiter_commit; this script runs where .git is)
for c in cfg2 cfg3; do f=$dst/${tag}_${c}_pmc_sq.txt; [ -f $f ] && ! grep -q '^commit:' $f && sed -i "1i commit: $(git rev-parse --short HEAD)" $f; done
cp $src/tile_trace.txt $dst/${tag}_tile_trace_h32_backward.txt 2>/dev/null; cp $src/micro_glds_offset.txt $dst/${tag}_micro_glds_offset.txt 2>/dev/null
cp $src/step_timeline_cfg2.txt $dst/${tag}_step_timeline_cfg2.txt 2>/dev/null; cp $src/step_timeline_cfg3.txt $dst/${tag}_step_timeline_cfg3.txt 2>/dev/null
cp $src/launch_origins_cfg2.txt $dst/${tag}_launch_origins_cfg2.txt 2>/dev/null; cp $src/bench_cfg2_one_rank_rccl_ll.json $dst/${tag}_bench_cfg2_one_rank_rccl_ll.json 2>/dev/null
