#!/bin/bash
# Round-3 A/B of the f16x2 kernels against the bf16x3 ones on one device (interleaved, same process settings).
# usage: tools/ab_r3.sh <outdir>
out=$1; mkdir -p $out
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS > $out/$name.json 2> $out/$name.err; python3 - "$out/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    k=d['roofline'].get('kernel_ms_per_step',{})
    print(f"{sys.argv[2]:34s} {d['value']:9.1f} {d['unit']:9s} {d['ms_per_step']:7.3f} ms/step  {k}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for rep in 1 2; do
ARGS=""
run cfg2_f16_$rep A=1
run cfg2_bf16_$rep PVS_BWD32=bf16 PVS_EGNN_F16X2=0
run cfg2_f16bwd_only_$rep PVS_EGNN_F16X2=0
run cfg2_f16fwd_only_$rep PVS_BWD32=bf16
done
ARGS="--config cfg3"
run cfg3_f16 A=1
run cfg3_bf16 PVS_EGNN_F16X2=0
ARGS="--config cfg5 --steps 300"
run cfg5_f16 A=1
run cfg5_bf16 PVS_EGNN_F16X2=0
ARGS="--infer"
run cfg2infer_f16 A=1
run cfg2infer_bf16 PVS_EGNN_F16X2=0
