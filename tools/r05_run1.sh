set -x
out=gpurun_out/r05_run1; mkdir -p $out
PVS_EGNN_LIB=$PWD/pointvs_amd/libpvs_egnn_r04head.so timeout 900 python -m pytest tests/test_gpu_lazy_scales.py -q -m gpu -x --no-header -p no:cacheprovider --co -q > /dev/null 2>&1
PVS_EGNN_LIB=$PWD/pointvs_amd/libpvs_egnn_r04head.so timeout 900 python -m pytest tests/test_gpu_lazy_scales.py -q -m gpu 2>&1 | tail -40 > $out/lazy_tests_r04head.txt
timeout 1200 python -m pytest tests/test_gpu_lazy_scales.py tests/test_gpu_properties.py -q -m gpu -k "lazy or tile_magnitudes or bias_sums or dynamic_range or binades" 2>&1 | tail -40 > $out/lazy_tests_new.txt
AB_ROUNDS=3 python tools/ab.py head=pointvs_amd/libpvs_egnn_r04head.so new=pointvs_amd/libpvs_egnn.so > $out/ab_head_vs_new.txt 2>&1
cat $out/lazy_tests_r04head.txt | tail -15; cat $out/lazy_tests_new.txt | tail -15; cat $out/ab_head_vs_new.txt
