export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_gputests_a.log
mkdir -p gpurun_out/r04_cfg3_split gpurun_out/r04_cfg3_def
PVS_EGNN_SPLIT_SMALL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_cfg3_split -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cfg3 > gpurun_out/r04_cfg3_split.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_cfg3_def -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --config cfg3 > gpurun_out/r04_cfg3_def.json 2>/dev/null
find gpurun_out/r04_cfg3_split gpurun_out/r04_cfg3_def -name '*kernel_trace.csv' -delete
for d in split def; do f=$(ls -t gpurun_out/r04_cfg3_$d/*/*kernel_stats.csv | head -1); python3 tools/kstats.py $f 7 14 > gpurun_out/r04_cfg3_$d.kstats.txt; done
