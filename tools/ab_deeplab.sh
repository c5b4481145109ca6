#!/bin/bash
# A/B on ONE box for the DeepLab per-launch path: <label>=<library>[,ENV=VAL...] ...   (alternating, R rounds)
#   tools/ab_deeplab.sh 2 old=backscrub_amd/libbsx_A.so new=backscrub_amd/libbsx.so three=backscrub_amd/libbsx_dbg.so,BSX_NO_CHAIN3=1
# prints per run: label, fps, ms per step, and the launches whose name matches $AB_GREP (default: the ASPP head)
R=$1; shift
for i in $(seq $R); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""
    if [ "$rest" != "$lib" ]; then envs=$(echo ${rest#*,} | tr ',' ' '); fi
    env $envs BSX_LIBRARY=$PWD/$lib python bench.py --model deeplab --batch 1024 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 3 --dump-launches gpurun_out/launches_$label.txt 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$label', round(d['value']), d['ms_per_step'], d.get('parity_sample'))"
    grep -E "${AB_GREP:-conv#64|conv#66-pool|conv#67}" gpurun_out/launches_$label.txt | grep " us " | sed 's/^/    /'
  done
done
