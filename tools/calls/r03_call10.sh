#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
run() { env "$@" python bench.py --no-extra-configs --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches'][:5]])"; }
for rep in 1 2; do
run X=1
run BSX_PLAN_NO_TOPDOWN=1
run BSX_RTC_NO_EARLY_FC=1
run BSX_PLAN_NO_TOPDOWN=1 BSX_RTC_NO_EARLY_FC=1
run BSX_NO_RTC=1
done
