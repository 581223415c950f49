#!/bin/bash
# The per-rank step times a 1 / 2 / 4 / 8-GPU strong-scaling run of BASELINE config 4 (global batch 256) would see,
# measured on ONE GPU: 256, 128, 64, 32 graphs per step. No scaling curve: a prediction's inputs.
# usage (GPU box): tools/strong_scaling_legs.sh > gpurun_out/<tag>_strong_scaling_one_gpu_legs.txt
for b in 256 128 64 32; do
  python3 bench.py --gpus 1 --batch $b --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['kernel_ms_per_step']
print(f'graphs per GPU {d[\"config\"][\"graphs_per_gpu\"]:4d}  {d[\"ms_per_step\"]:8.3f} ms/step  {d[\"value\"]:8.1f} graphs/s  per graph {d[\"ms_per_step\"]/d[\"config\"][\"graphs_per_gpu\"]*1e3:7.1f} us  kernels {k}')"
done
