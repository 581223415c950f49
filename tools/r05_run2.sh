out=gpurun_out/r05_run2; mkdir -p $out
python tools/dynamic_range_probe.py > $out/dynamic_range.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $out/gpu_suite.txt
tail -5 $out/gpu_suite.txt
