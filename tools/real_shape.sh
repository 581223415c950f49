#!/bin/bash
# The launch-bound regime (NOT a BASELINE shape): bench.py --config real4A, eager against the step replayed from a hipGraph,
# at several batch sizes.   usage: tools/real_shape.sh [outfile]
out=${1:-gpurun_out/real_shape.txt}; : > $out
for b in 32 8; do for gm in 0 1; do
  python3 bench.py --config real4A --batch $b --steps 200 --warmup 30 --graph $gm --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print(f\"batch $b  {'hipGraph replay' if $gm else 'eager          '}  {d['value']:9.1f} graphs/s  {d['ms_per_step']:6.3f} ms/step   kernels (eager, ms/step): fwd {k['edge_fwd']:.3f} bwd {k['edge_bwd']:.3f} gather {k['col_gather']:.3f} prepare {k['graph_prepare']:.3f}\")" >> $out
done; done
cat $out
