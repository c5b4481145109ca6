#!/bin/bash
# round 5, call 3: (a) seg parity after the head went back to the dense tile, (b) same-box A/B old / new / new at 4 tail workgroups per CU, (c) the N > 1 path of the
# rewritten bench.py on this 1-GPU box (two ranks sharing the GPU, gloo counters): must print one compact line and exit 0
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05c_pytest.txt
grep -q "failed\|error" gpurun_out/r05c_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { ( cd $1; env $4 timeout 300 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 8 --steps 100 --warmup 10 --ramp-seconds 1.0 $3 --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('/tmp/ab_detail.json')); t={x['name']:x['ms'] for x in f['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'net', f['stage_ms']['network'], ' '.join('%s=%.4f' % (k, t[k]) for k in sorted(t)))" ); }
for cfg in "--model lite" "--model mlkit --width 1280 --height 720" "--model full --width 1280 --height 720 --batch 1024"; do
  for i in 1 2; do run $ROOT/_ab_old old "$cfg" X=1; run $ROOT new "$cfg" X=1; run $ROOT new-tail4wg "$cfg" BSX_SEG_TAIL_WGS=4; done
done 2>&1 | tee gpurun_out/r05c_seg_ab.txt
S=$(date +%s)
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --ramp-seconds 0.5 --cpu-seconds 3 > gpurun_out/r05c_bench_gpus2_on_one_gpu.json 2> gpurun_out/r05c_bench_gpus2.err
echo "gpus2 rc=$? wall=$(( $(date +%s) - S )) s bytes=$(wc -c < gpurun_out/r05c_bench_gpus2_on_one_gpu.json)" | tee gpurun_out/r05c_bench_gpus2.txt
tail -5 gpurun_out/r05c_bench_gpus2.err; head -c 1500 gpurun_out/r05c_bench_gpus2_on_one_gpu.json
