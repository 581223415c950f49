#!/bin/bash
# Sustained-clock evidence: 1000 timed steps per configuration (6 s / 20 s of back-to-back
# kernels instead of the default bench's 0.12 s) with the shader / memory clocks sampled from rocm-smi twice a second
# while the loop runs.   usage (GPU box): tools/sustained.sh <tag>   -> gpurun_out/<tag>_sustained_{cfg2,cfg3}.json
tag=$1
for cfg in cfg2 cfg3; do
  clk=gpurun_out/${tag}_sustained_${cfg}_clocks.txt
  : > $clk
  ( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | tr -s ' ' | tr '\n' ';' >> $clk; echo >> $clk; sleep 0.5; done ) &
  sampler=$!
  python3 bench.py --config $cfg --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/${tag}_sustained_${cfg}.json 2>/dev/null
  kill $sampler
  python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_short_${cfg}.json 2>/dev/null
  python3 - $cfg $clk gpurun_out/${tag}_sustained_${cfg}.json gpurun_out/${tag}_short_${cfg}.json <<'PY'
import json, re, sys
cfg, clk, longf, shortf = sys.argv[1:]
s = [int(m) for m in re.findall(r'sclk[^;]*?\((\d+)Mhz\)', open(clk).read())]
rd = lambda f: json.loads([l for l in open(f) if l.startswith('{')][-1])
a, b = rd(longf), rd(shortf)
mid = s[len(s) // 4: -len(s) // 8 or None] or s
print(f'{cfg}: 1000 steps {a["value"]:.1f} graphs/s ({a["ms_per_step"]} ms/step), 20 steps {b["value"]:.1f} ({b["ms_per_step"]}); '
      f'sclk samples {len(s)}: min {min(s) if s else None} median-of-loop {sorted(mid)[len(mid) // 2] if mid else None} max {max(s) if s else None} MHz')
PY
done
