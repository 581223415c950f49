export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -m gpu -k "edgeres or all_on or ragged or kernel_families or bitwise" 2>&1 | tail -5 > gpurun_out/r04_tests_c.log
for fl in "edge_residual=True,rezero=True" "edge_residual=True,gated_residual=True" "edge_residual=True,edge_attention=True,rezero=True" "edge_residual=True,edge_attention=True,gated_residual=True"; do
  for lib in oldf16 new; do
    L=""; [ $lib = oldf16 ] && L="PVS_EGNN_LIB=$PWD/pointvs_amd/libpvs_egnn_oldf16.so"
    line=$(env $L python3 bench.py --config cfg2 --steps 8 --warmup 3 --no-cpu-baseline --model-flags "$fl" 2>/dev/null | grep '^{' | tail -1)
    python3 -c "
import json,sys
d=json.loads(sys.argv[2]); r=d['roofline']
print(f'{sys.argv[1]:70s} {d[\"value\"]:8.1f} graphs/s  edge backward {r[\"avg_launch_ms\"]:.3f} ms/launch')" "$lib cfg2 $fl" "$line"
  done
done > gpurun_out/r04_variants_gated_rezero.txt
