#!/bin/bash
# Per-launch times of the non-default instantiations of the H = 32 edge backward (NOT BASELINE configurations):
# cfg2 + one model flag, round 3's f16x2 kernel against round 2's dispatch (PVS_BWD32=bf16: six-term bf16 kernel,
# which hands edge residual + attention to the round-1 kernel).   usage: tools/variants_r3.sh <outfile>
out=$1
run() { name=$1; flags=$2; shift 2; line=$(env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --model-flags "$flags" 2>/dev/null | grep '^{' | tail -1);
  python3 - "$name" "$line" <<'PY' >> $out
import json,sys
d=json.loads(sys.argv[2]); r=d['roofline']
print(f"{sys.argv[1]:44s} {d['value']:8.1f} graphs/s  {d['ms_per_step']:7.3f} ms/step  edge backward {r['avg_launch_ms']:.3f} ms/launch  ({r['kernel'].split()[0]})")
PY
}
: > $out
for fl in "edge_attention=True" "edge_residual=True" "edge_residual=True,edge_attention=True" "edge_attention=True,softmax_attention=True,node_attention=True"; do
  run "f16x2  $fl" "$fl" A=1
  run "round2 $fl" "$fl" PVS_BWD32=bf16
done
cat $out
