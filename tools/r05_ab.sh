# usage: tools/r05_ab.sh <tag> [AB_ARGS...]   A/B of the round-4 library against the current build + the fast parity subset
tag=$1; shift
out=gpurun_out/r05_$tag; mkdir -p $out
AB_ROUNDS=${AB_ROUNDS:-3} AB_ARGS="$*" python tools/ab.py head=pointvs_amd/libpvs_egnn_r04head.so new=pointvs_amd/libpvs_egnn.so > $out/ab.txt 2>&1
cat $out/ab.txt
timeout 1500 python -m pytest tests/test_gpu_lazy_scales.py tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py -q -m gpu -x 2>&1 | tail -5 > $out/tests.txt
cat $out/tests.txt
