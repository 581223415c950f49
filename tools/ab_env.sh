#!/bin/bash
# On the GPU box: interleaved A/B of an environment switch on bench.py.  usage: tools/ab_env.sh VAR=VALUE [rounds] [bench args...]
kv=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do
  python3 bench.py --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base      ', d['value'], d['ms_per_step'])"
  env $kv python3 bench.py --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['value'], d['ms_per_step'])"
done
