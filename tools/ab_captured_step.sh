#!/bin/bash
# On the GPU box: eager step against the whole step replayed from a hipGraph (bench.py --graph 1), BASELINE training
# shapes, interleaved rounds on one device.   usage: tools/ab_captured_step.sh [rounds] [steps]
rounds=${1:-3}; steps=${2:-20}
out=gpurun_out/ab_captured_step.txt
mkdir -p gpurun_out
: > $out
for cfg in cfg2 cfg3; do
  for r in $(seq 1 $rounds); do
    for g in 0 1; do
      line=$(python3 bench.py --config $cfg --graph $g --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
      python3 - "$cfg" "$r" "$g" "$line" >> $out <<'PY'
import json, sys
cfg, r, g, line = sys.argv[1:5]
try:
    d = json.loads(line)
    k = d['roofline'].get('kernel_ms_per_step', {})
    print(f"{cfg} round {r} graph={g} launch={d['config'].get('launch')!r:40s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}  "
          f"kernels fwd {k.get('edge_fwd')} bwd {k.get('edge_bwd')} col {k.get('col_gather')} prep {k.get('graph_prepare')}  final_loss {d['config'].get('final_loss')}")
except Exception as exc:
    print(f'{cfg} round {r} graph={g} FAILED {exc!r} {line[:200]}')
PY
    done
  done
done
cat $out
