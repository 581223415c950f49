#!/bin/bash
# On the GPU box: per-kernel average durations (rocprofv3 --stats) of bench.py under several library builds.
# usage: tools/variant_kernel_times.sh "<bench args>" "<kernel name pattern>" name1 name2 ...   (name "base" = the shipped library)
args=$1; pat=$2; shift; shift
export TMPDIR=/tmp
for name in "$@"; do
  lib=$PWD/pointvs_amd/libpvs_egnn_$name.so; [ "$name" = base ] && lib=$PWD/pointvs_amd/libpvs_egnn.so
  out=gpurun_out/vk_$name; rm -rf $out; mkdir -p $out
  PVS_EGNN_LIB=$lib rocprofv3 --kernel-trace --stats -d $out -o v --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $args > $out/bench.json 2> $out/err.txt
  rm -f $out/*kernel_trace.csv
  echo "== $name: $(grep -o '"value": [0-9.]*' $out/bench.json | head -1)"
  python3 - "$out/v_kernel_stats.csv" "$pat" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        print('   %-60s calls %5s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
