#!/bin/bash
# A/B two builds of libbsx.so on ONE box (boxes of the pool differ by +-5 %): alternating runs of the short bench
#   tools/ab_bench.sh backscrub_amd/libbsx_A.so backscrub_amd/libbsx.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    BSX_LIBRARY=$PWD/$L python bench.py --no-extra-configs --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); fp=[t['ms'] for t in d['top_launches'] if t['name']=='frame_program']
print('$L', round(d['value']), d['ms_per_step'], 'frame_program', fp)"
  done
done
