#!/bin/bash
# On the GPU box: interleaved runs of bench.py under several environment settings ("-" = none).
# usage: tools/ab_env_multi.sh <rounds> "<bench args>" SETTING [SETTING...]      (SETTING: VAR=VALUE or -)
rounds=$1; args=$2; shift; shift
for r in $(seq $rounds); do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then e=""; else e="$kv"; fi
    env $e python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$kv', d['value'], d['ms_per_step'])"
  done
done
