#!/bin/bash
# round 4, call 23: the N > 1 job end to end on the final tree (two ranks sharing the box's one GPU: a plumbing run) — self-launched and through torch.distributed.run as the driver does
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 500 python bench.py --gpus 2 --steps 20 --warmup 3 --cpu-seconds 4 > gpurun_out/r04u_bench_n2_self.json 2> gpurun_out/r04u_bench_n2_self.err; echo "rc=$?"; tail -c 500 gpurun_out/r04u_bench_n2_self.err
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --cpu-seconds 4 > gpurun_out/r04u_bench_n2_torchrun.json 2> gpurun_out/r04u_bench_n2_torchrun.err; echo "rc=$?"; tail -c 500 gpurun_out/r04u_bench_n2_torchrun.err
python - <<'P'
import json
for f in ("gpurun_out/r04u_bench_n2_self.json", "gpurun_out/r04u_bench_n2_torchrun.json"):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    print(f, len(lines), "JSON line(s)")
    d = json.loads(lines[-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "steps", "ms_per_step", "scaling")}, d.get("collective"), list(d.keys())[:40])
    for k in ("configs1", "configs4"):
        c = d.get(k) or {}
        print(k, {q: c.get(q) for q in ("value", "ms_per_step", "per_rank_fps", "rank0_alone_fps", "efficiency_vs_rank0_alone")})
    print("cpu_baseline" in d, d.get("second_device_check"), d.get("gpu_sharing"))
P
