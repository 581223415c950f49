out=gpurun_out/r05_run3; mkdir -p $out
AB_ROUNDS=3 python tools/ab.py head=pointvs_amd/libpvs_egnn_r04head.so swz=pointvs_amd/libpvs_egnn.so > $out/ab_swizzle.txt 2>&1; cat $out/ab_swizzle.txt
AB_ROUNDS=2 python tools/ab.py base=pointvs_amd/libpvs_egnn.so nogz1=pointvs_amd/libpvs_egnn_nogz1.so ngres=pointvs_amd/libpvs_egnn_ngres.so both=pointvs_amd/libpvs_egnn_colceil.so > $out/ab_column_ceiling.txt 2>&1; cat $out/ab_column_ceiling.txt
tools/micro/valu_issue_bench.bin sigmoid > $out/micro_sigmoid.txt 2>&1; cat $out/micro_sigmoid.txt
timeout 900 python -m pytest tests/test_gpu_lazy_scales.py tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py -q -m gpu -x 2>&1 | tail -3
bash tools/pmc_sq.sh r05swz > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_r05swz > $out/pmc_sq_summary.txt 2>&1; python3 tools/pmc_header.py $out/pmc_sq_summary.txt cfg2 | head -8
