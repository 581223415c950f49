out=gpurun_out/r05_small; mkdir -p $out
export TMPDIR=/tmp
line() { grep '^{' | tail -1; }
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $out/suite.txt; cat $out/suite.txt
bash tools/real_shape.sh $out/real_shape.txt > /dev/null
PVS_EDGES_PER_WAVE=512 bash tools/real_shape.sh $out/real_shape_512_edges_per_wave.txt > /dev/null
python3 tools/train_capture_bench.py 2>&1 | grep "^capture" > $out/train_capture.txt
python3 bench.py --config real4A --steps 200 --warmup 20 --graph 0 --no-cpu-baseline 2>/dev/null | line > $out/real_shape_eager.json
python3 bench.py --config real4A --steps 200 --warmup 20 --graph 1 --no-cpu-baseline 2>/dev/null | line > $out/real_shape_graph.json
rocprofv3 --kernel-trace --output-format csv -d $out/trace_real4A -- python3 bench.py --config real4A --steps 3 --warmup 2 --graph 0 --no-cpu-baseline > /dev/null 2>&1
python3 tools/step_timeline.py $out/trace_real4A > $out/step_timeline_real4A.txt 2>&1
find $out/trace_real4A -name '*.csv' -delete
python3 bench.py --steps 20 --warmup 5 2>/dev/null | line > $out/bench_cfg2.json
cat $out/real_shape.txt $out/real_shape_512_edges_per_wave.txt $out/train_capture.txt; tail -2 $out/step_timeline_real4A.txt
python3 -c "
import json
d=json.loads(open('$out/bench_cfg2.json').read()); print('cfg2', d['value'], d['ms_per_step'], {k:(v['value'],v['ms_per_step']) for k,v in d['secondary'].items()})"
