#!/bin/bash
# XCD-aware (frame, tile) order of the tile kernels: whole GPU test suite, then A/B against the previous build is not possible in one tree -> measure + PMC fetch of the default job
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03n_pytest.txt 2>&1; tail -4 gpurun_out/r03n_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches']])"; }
for rep in 1 2; do
run BSX_XCD_TILES=1 -- --steps 200 --warmup 20 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --steps 200 --warmup 20 --ramp-seconds 1
done
run BSX_XCD_TILES=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_XCD_TILES=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for v in 1 0; do
BSX_XCD_TILES=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_r03n_$v -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs --profile-iters 1 --ramp-seconds 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py --pmc $R/gpurun_out/pmc_fetch_r03n_$v/bench_results.db | grep -v "at::\|rocclr" > $R/gpurun_out/r03n_fetch_xcd$v.md; rm -rf $R/gpurun_out/pmc_fetch_r03n_$v
done
cat $R/gpurun_out/r03n_fetch_xcd1.md $R/gpurun_out/r03n_fetch_xcd0.md
