#!/bin/bash
# On the GPU box: by-column placement through LDS-sorted tiles (default) against the direct scatter (PVS_CSC_TILES=0).
mkdir -p gpurun_out; out=gpurun_out/ab_csc_tiles.txt; : > $out
for cfg in cfg2 cfg3 real4A; do
  for r in 1 2 3; do
    for v in 1 0; do
      PVS_CSC_TILES=$v python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline'].get('kernel_ms_per_step',{})
print('$cfg round $r tiles=$v  ms_per_step %.3f  value %.1f  graph_prepare %s' % (d['ms_per_step'], d['value'], k.get('graph_prepare')))" >> $out
    done
  done
done
cat $out
