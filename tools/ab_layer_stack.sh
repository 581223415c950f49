#!/bin/bash
# On the GPU box: one autograd node + one C call per EGNN layer (PVS_EGNN_STACK=0) against the layer loop as one call each
# way (pvs_egnn_stack_fwd / _bwd, the default), interleaved rounds on one device: the launch-bound shape (real4A = the
# reference's CLI defaults, batches of 32 and 8; eager and replayed from a hipGraph) and the BASELINE training shapes.
# usage: tools/ab_layer_stack.sh [rounds]
rounds=${1:-3}
out=gpurun_out/ab_layer_stack.txt
mkdir -p gpurun_out
: > $out
run() {   # label, bench args...
  label=$1; shift
  for r in $(seq 1 $rounds); do
    for m in 0 1; do
      line=$(PVS_EGNN_STACK=$m python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
      python3 - "$label" "$r" "$m" "$line" >> $out <<'PY'
import json, sys
label, r, m, line = sys.argv[1:5]
try:
    d = json.loads(line)
    print(f"{label:28s} round {r} PVS_EGNN_STACK={m}  {d['ms_per_step']:7.3f} ms/step  {d['value']:9.1f} {d['unit']}  final_loss {d['config'].get('final_loss')}")
except Exception as exc:
    print(f'{label} round {r} stack={m} FAILED {exc!r} {line[:200]}')
PY
    done
  done
}
run 'real4A batch 32 eager'  --config real4A --batch 32 --steps 200 --warmup 30 --graph 0
run 'real4A batch 8 eager'   --config real4A --batch 8 --steps 200 --warmup 30 --graph 0
run 'real4A batch 32 replay' --config real4A --batch 32 --steps 200 --warmup 30 --graph 1
run 'cfg2'                   --config cfg2 --steps 20 --warmup 5
run 'cfg3'                   --config cfg3 --steps 10 --warmup 3
cat $out
