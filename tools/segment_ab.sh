#!/bin/bash
# Would cache-sized segments pay? The backward writes 144 B per edge of column-side records
# (g_z1 row + 16-byte record) and the column gather reads them back; a segment of <= 4 graphs (<= 190 MB) would keep
# them inside the 256 MB Infinity Cache. A segment pipeline IS a sequence of small-batch launches, so its kernels cost
# what the same kernels cost on a batch of that size: measured here under hipGraph replay (no host launch cost),
# per graph.   usage: tools/segment_ab.sh <outfile>
out=$1; : > $out
for b in 32 16 8 4 2 1; do
  steps=$(( 640 / b )); [ $steps -gt 120 ] && steps=120
  line=$(python3 bench.py --batch $b --graph 1 --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
  python3 - "$b" "$line" <<'PY' >> $out
import json,sys
b=int(sys.argv[1]); d=json.loads(sys.argv[2]); k=d['roofline']['kernel_ms_per_step']
print(f"graphs per launch {b:3d}: step {d['ms_per_step']:7.3f} ms = {d['ms_per_step']/b*1e3:7.1f} us/graph | per graph: edge bwd {k['edge_bwd']/b*1e3:6.1f} us  col gather {k['col_gather']/b*1e3:6.1f} us  bwd+gather {(k['edge_bwd']+k['col_gather'])/b*1e3:6.1f} us  edge fwd {k['edge_fwd']/b*1e3:6.1f} us  prepare {k['graph_prepare']/b*1e3:6.1f} us")
PY
done
cat $out
