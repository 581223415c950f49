out=gpurun_out/r05_run5; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lazy_scales.py tests/test_gpu_properties.py -q -m gpu -x -k "not k64 and not wide and not h64" 2>&1 | tail -6
bash tools/variants_r3b.sh $out/variants_pre4.txt $PWD/pointvs_amd/libpvs_egnn_pre4.so > /dev/null 2>&1
bash tools/variants_r3b.sh $out/variants_new.txt > /dev/null 2>&1
echo PRE; cat $out/variants_pre4.txt; echo NEW; cat $out/variants_new.txt
