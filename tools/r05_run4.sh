out=gpurun_out/r05_run4; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py -q -m gpu -k "k64 or cfg3 or c3_ or h64" 2>&1 | tail -3
AB_ROUNDS=3 AB_ARGS="--config cfg3" python tools/ab.py base=pointvs_amd/libpvs_egnn_base.so new=pointvs_amd/libpvs_egnn.so > $out/ab_cfg3.txt 2>&1; cat $out/ab_cfg3.txt
bash tools/pmc_sq.sh r05cfg3 --config cfg3 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_r05cfg3 > $out/pmc_sq_cfg3_summary.txt 2>&1; python3 tools/pmc_header.py $out/pmc_sq_cfg3_summary.txt cfg3 | grep "k_edge_bwd_h64\|k_edge_fwd" | cut -c1-420
