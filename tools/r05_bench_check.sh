# the driver's invocation (with the secondary configurations) + the launch-bound shape, eager and captured
out=gpurun_out/r05_bench; mkdir -p $out
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/default.json 2> $out/default.err
tail -4 $out/default.err
python bench.py --config real4A --steps 50 --warmup 10 --graph 0 --no-cpu-baseline > $out/real4A_eager.json 2> $out/real4A_eager.err
python bench.py --config real4A --steps 50 --warmup 10 --graph 1 --no-cpu-baseline > $out/real4A_graph.json 2> $out/real4A_graph.err
python - <<'PY'
import json
for f in ('default','real4A_eager','real4A_graph'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r05_bench/{f}.json') if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('avg_launch_ms_full_work'), d['roofline'].get('frac'), d['roofline'].get('frac_full_work'), d['config'].get('launch'))
        for k,v in d.get('secondary',{}).items(): print('   ', k, v['value'], v['ms_per_step'], v['roofline']['frac'], v['roofline']['avg_launch_ms'])
    except Exception as e:
        print(f, 'FAILED', e); print(open(f'gpurun_out/r05_bench/{f}.err').read()[-1500:])
PY
