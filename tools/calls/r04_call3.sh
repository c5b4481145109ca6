#!/bin/bash
# round 4, call 3: (1) the N > 1 job end to end, two ranks sharing the box's one GPU (call 2 found a deadlock there: fixed); (2) mixed read/write HBM ceiling;
# (3) per-op wait/body split of the specialised middle kernel; (4) g1 upper bound: the middle kernel issuing a quarter of its 1x1 MFMAs (timing only)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 420 python bench.py --gpus 2 --steps 40 --warmup 5 --cpu-seconds 6 > gpurun_out/r04_call3_bench_n2.json 2> gpurun_out/r04_call3_bench_n2.err; echo "n2 rc=$?"; tail -c 800 gpurun_out/r04_call3_bench_n2.err; head -c 2500 gpurun_out/r04_call3_bench_n2.json; echo
timeout 120 tools/microbench_mix 2>&1 | tee gpurun_out/r04_microbench_mix.txt
BSX_RTC_FINE=1 timeout 200 python tools/program_timeline.py lite 256 --fine > gpurun_out/r04_timeline_lite_fine.txt 2>&1; tail -45 gpurun_out/r04_timeline_lite_fine.txt
run() { env $2 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --no-host-io --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1', '$3', d['ms_per_step'], 'frame_program', t.get('frame_program'))"; }
for i in 1 2; do
  run default "" ""
  run mfma_quarter "BSX_RTC_EXP_MFMA=1" ""
done 2>&1 | tee gpurun_out/r04_g1_upper_bound.txt
for i in 1 2; do
  run default "" "--model mlkit --width 1280 --height 720"
  run mfma_quarter "BSX_RTC_EXP_MFMA=1" "--model mlkit --width 1280 --height 720"
done 2>&1 | tee -a gpurun_out/r04_g1_upper_bound.txt
