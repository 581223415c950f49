#!/bin/bash
# Per-launch times of the non-default instantiations of the edge backward (NOT BASELINE configurations): cfg2 (H = 32)
# and cfg3 (H = 64) + model flags, one library build.   usage: tools/variants_r3b.sh <outfile> [lib.so]
out=$1; lib=$2
run() { cfg=$1; flags=$2; line=$(env ${lib:+PVS_EGNN_LIB=$lib} python3 bench.py --config $cfg --steps 8 --warmup 3 --no-cpu-baseline --model-flags "$flags" 2>/dev/null | grep '^{' | tail -1);
  python3 - "$cfg $flags" "$line" <<'PY' >> $out
import json,sys
d=json.loads(sys.argv[2]); r=d['roofline']
print(f"{sys.argv[1]:64s} {d['value']:8.1f} graphs/s  {d['ms_per_step']:7.3f} ms/step  edge backward {r['avg_launch_ms']:.3f} ms/launch  ({r['kernel'].split()[0]})")
PY
}
: > $out
for fl in "edge_attention=True" "edge_residual=True" "edge_residual=True,rezero=True" "edge_residual=True,edge_attention=True" "edge_residual=True,edge_attention=True,gated_residual=True" "edge_attention=True,softmax_attention=True,node_attention=True"; do
  run cfg2 "$fl"
done
run cfg3 "edge_residual=True"
run cfg3 "edge_residual=True,gated_residual=True"
cat $out
