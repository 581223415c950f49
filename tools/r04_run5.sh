export TMPDIR=/tmp
python -m pytest tests/test_gpu_training_trajectory.py tests/test_gpu_properties.py -q -m gpu -s -k "trajectory or broken_layout or dynamic_range or folded or prepare_by or by_column" 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r04_tests_b.log
mkdir -p gpurun_out/r04_cfg3_b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_cfg3_b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cfg3 > gpurun_out/r04_cfg3_b.json 2>/dev/null
find gpurun_out/r04_cfg3_b -name '*kernel_trace.csv' -delete
f=$(ls -t gpurun_out/r04_cfg3_b/*/*kernel_stats.csv | head -1); python3 tools/kstats.py $f 7 12 > gpurun_out/r04_cfg3_b.kstats.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_cfg2_b.json 2>/dev/null
python3 bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r04_cfg3_b2.json 2>/dev/null
