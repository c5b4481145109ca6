"""N>1 path on CPU: world_size-2 gloo processes exercising the same sharding + counter all-reduce
bench.py uses on RCCL (the compute itself has no CPU path and is not run here)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from backscrub_amd.dist import reduce_counters, shard_streams


def test_shard_streams_partitions_exactly():
    for total in (1, 7, 256, 8192, 8195):
        for world in (1, 2, 3, 8):
            blocks = [shard_streams(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, end = shard_streams(513, world, rank)
    frames = (end - start) * 10
    elapsed = 1.0 + 0.5 * rank
    out = reduce_counters(frames, elapsed, 1000 + rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_counter_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, (frames, elapsed, checksum) in res:
        assert frames == 5130.0 and elapsed == 1.5 and checksum == 2001


def _coll_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from backscrub_amd.dist import Collective
    c = Collective(gpu=False)
    c.barrier()
    got = (c.describe(), c.reduce(10 * (rank + 1), 1.0 + rank, 7), c.gather(100.0 + rank))
    c.close()
    q.put((rank, got))


def test_collective_object_world2():
    """backscrub_amd.dist.Collective — what bench.py holds at N > 1 — on gloo: probe (ranks_seen), barrier, reduce, gather, close."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_coll_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        desc, red, gat = res[r]
        assert desc == {"backend": "gloo", "ranks_seen": 2, "world_size": 2}
        assert red == (30.0, 2.0, 14) and gat == [100.0, 101.0]


def test_collective_is_a_no_op_for_one_rank():
    from backscrub_amd.dist import Collective
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    try:
        c = Collective(gpu=True)          # world 1: nothing is initialised, no GPU is touched
        c.barrier()
        assert c.ranks_seen == 1 and c.reduce(5, 2.0, 7) == (5.0, 2.0, 7) and c.gather(3.0) == [3.0]
        c.close()
    finally:
        os.environ.update(env)


def test_reduce_counters_without_group():
    assert reduce_counters(5, 2.0, 7) == (5.0, 2.0, 7)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it must spawn its own ranks (torch.distributed.run, 127.0.0.1) —
    the plumbing leg (--selftest-dist: gloo, no GPU work) goes through exactly that launch, rendezvous and counter reduction."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-dist", "--batch", "8", "--steps", "5", "--cpu-seconds", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["frames"] == 80.0 and abs(d["elapsed_max"] - 0.00625) < 1e-9 and d["checksum"] == 2001
    assert d["metric"] == json.load(open(os.path.join(root, "BASELINE.json")))["metric"]
    # the N > 1 shape of the line (bench.multi_gpu_sections, shared with the GPU run): what the collective really connected, per-rank rates,
    # the north-star job's slice on every rank, both against rank 0 alone, and the CPU baseline kept
    assert d["ranks_seen"] == 2 and d["collective"] == {"backend": "gloo", "ranks_seen": 2, "world_size": 2}
    c1, c4 = d["configs1"], d["configs4"]
    assert len(c1["per_rank_fps"]) == 2 and c1["per_rank_fps_min"] == min(c1["per_rank_fps"]) and c1["per_rank_fps_max"] == max(c1["per_rank_fps"])
    assert abs(c1["value"] - d["value"]) < 1e-6 and abs(c1["efficiency_vs_rank0_alone"] - 0.8) < 1e-9       # rank 1 is 25 % slower by construction; time = max over ranks
    assert c4["streams_total"] == 2048 and "segm_full_v679" in c4["workload"] and "1280x720" in c4["workload"] and len(c4["per_rank_fps"]) == 2
    assert abs(c4["value"] - 2 * 1024 * 1e3 / c4["ms_per_step"]) / c4["value"] < 1e-6
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["host"]["model"] and d["host"]["logical_cpus"] >= 1


def test_bench_walks_the_north_star_rank_count_world8():
    """BASELINE configs[4] is an 8-rank job (8192 HD streams / 8 GPUs).  No node with more than one GPU existed in any round, so the exact rank count is walked
    here on gloo: `bench.py --gpus 8 --selftest-dist` self-launches 8 ranks through torch.distributed.run on 127.0.0.1, every rank takes the same order of solo
    runs, barriers and reductions as the GPU job, and rank 0 prints the line: 8 ranks seen by the collective, 8192 streams in the configs[4] leg, per-rank rates
    from the gather, time = max over ranks (the slowest rank — 2.75x by construction — sets the job's rate)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--selftest-dist", "--batch", "256", "--steps", "4", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["collective"] == {"backend": "gloo", "ranks_seen": 8, "world_size": 8}
    assert d["frames"] == 8 * 256 * 4 and d["checksum"] == sum(1000 + r for r in range(8))
    assert abs(d["elapsed_max"] - 4 * 1e-3 * (1.0 + 0.25 * 7)) < 1e-9                 # the slowest rank's time
    c1, c4 = d["configs1"], d["configs4"]
    assert len(c1["per_rank_fps"]) == 8 and c1["per_rank_fps"] == sorted(c1["per_rank_fps"], reverse=True)        # rank r is (1 + r/4)x slower, gathered in rank order
    assert abs(c1["efficiency_vs_rank0_alone"] - 1.0 / 2.75) < 1e-4                        # rounded to 4 places in the line
    assert c4["streams_total"] == 8192 and "segm_full_v679" in c4["workload"] and "1280x720" in c4["workload"] and len(c4["per_rank_fps"]) == 8
    assert abs(c4["value"] - 8 * 1024 * 1e3 / c4["ms_per_step"]) / c4["value"] < 1e-6


def test_ranks_bind_to_the_numa_node_of_their_gpu(tmp_path):
    """bench.py at N > 1: each rank restricts itself to the CPUs of its GPU's NUMA node (sysfs: bus/pci/devices/<id>/numa_node → devices/system/node/nodeK/cpulist),
    intersected with the affinity mask it was given; a platform that does not expose the topology (numa_node = -1, no sysfs entry) leaves the mask alone."""
    from backscrub_amd.dist import bind_to_gpu_numa, gpu_numa_cpus, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == [] and parse_cpulist("5") == [5]
    mine = sorted(os.sched_getaffinity(0))
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c5:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("%d,100000-100003\n" % mine[-1])           # one CPU we own + CPUs that do not exist here
    assert gpu_numa_cpus("0000:C5:00.0", str(tmp_path)) == (1, [mine[-1], 100000, 100001, 100002, 100003])
    rec = bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id="0000:c5:00.0", dry_run=True)
    assert rec == {"bound": False, "pci": "0000:c5:00.0", "node": 1, "cpus": 1} and sorted(os.sched_getaffinity(0)) == mine
    try:
        rec = bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id="0000:c5:00.0")
        assert rec["bound"] is True and sorted(os.sched_getaffinity(0)) == [mine[-1]]
    finally:
        os.sched_setaffinity(0, mine)
    (dev / "numa_node").write_text("-1\n")
    rec = bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id="0000:c5:00.0")
    assert rec["bound"] is False and "why" in rec and sorted(os.sched_getaffinity(0)) == mine
    assert bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id="0000:ff:00.0")["bound"] is False          # unknown device
    assert bind_to_gpu_numa(None, dry_run=True) == {"bound": False, "why": "no device"}
    (node / "cpulist").write_text("100000-100003\n")
    (dev / "numa_node").write_text("1\n")
    assert "affinity" in bind_to_gpu_numa(0, sysfs=str(tmp_path), pci_bus_id="0000:c5:00.0")["why"] and sorted(os.sched_getaffinity(0)) == mine
