out=gpurun_out/r05_run6; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
bash tools/variants_r3b.sh $out/variants.txt > /dev/null 2>&1; cat $out/variants.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
