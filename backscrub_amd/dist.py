"""Multi-GPU plumbing: streams are independent (per-stream state only, lib/libbackscrub.cc:46-48),
so the job shards contiguous blocks of streams across ranks with NO data-path collective.  The one
collective is the all-reduce of the throughput counters {frames (sum), elapsed (max), checksum (sum)}
— RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  SURVEY.md §8(e).

`Collective` is what bench.py talks to: one process group that carries BOTH backends ("cpu:gloo,cuda:nccl"), a probe that
proves RCCL works by actually all-reducing a device tensor (→ `ranks_seen`), and — if RCCL cannot be brought up on this node —
a labelled fall-back to the same counter reduction over gloo, so that a broken fabric library costs the report a label, not the
measurement."""
from __future__ import annotations

import datetime
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_streams(total_streams: int, world: int, rank: int):
    """Contiguous block of global stream ids owned by `rank` (first `total % world` ranks get one extra)."""
    base, extra = divmod(total_streams, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def reduce_counters(frames: float, elapsed: float, checksum: int, device=None):
    """→ (total frames, max elapsed over ranks, checksum sum mod 2^40).  Works with any initialised backend;
    with no process group it returns the local values."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(frames), float(elapsed), int(checksum) % (1 << 40)
    s = torch.tensor([float(frames), float(int(checksum) % (1 << 40))], dtype=torch.float64, device=device)
    m = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return s[0].item(), m[0].item(), int(s[1].item()) % (1 << 40)


class Collective:
    """The job's only communication: barriers around the timed region and the counter reduction after it.

    gpu=True : group "cpu:gloo,cuda:nccl"; the probe all-reduces a ones tensor ON THE GPU (RCCL) — `ranks_seen` is its result, so the JSON
               line says how many ranks RCCL really connected; every rank then agrees (over gloo) whether all probes succeeded; if any failed the
               job continues on CPU tensors over gloo and `backend` says why.
    gpu=False: plain gloo (CPU tests, `bench.py --selftest-dist`)."""

    def __init__(self, gpu: bool, timeout_s: int = 600):
        self.rank, self.world, self.local_rank = env_rank_world()
        self.backend, self.note, self.ranks_seen, self.device = "none", "", 1, None
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist
        to = datetime.timedelta(seconds=timeout_s)
        if not gpu:
            dist.init_process_group("gloo", timeout=to)
            self.backend, self.device = "gloo", torch.device("cpu")
        else:
            dist.init_process_group("cpu:gloo,cuda:nccl", timeout=to)          # RCCL communicators are created lazily, by the probe below
            ok, why = 1, ""
            try:
                t = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", self.local_rank))
                dist.all_reduce(t)
                torch.cuda.synchronize()
                if int(t.item()) != self.world:
                    ok, why = 0, "RCCL all-reduce of ones returned %r on rank %d" % (t.item(), self.rank)
            except Exception as e:  # noqa: BLE001 — any RCCL bring-up failure (IPC handles, missing fabric, version skew)
                ok, why = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)                         # CPU tensor → gloo: every rank takes the same branch
            if int(flag.item()) == 1:
                self.backend, self.device = "nccl (RCCL)", torch.device("cuda", self.local_rank)
            else:
                self.backend, self.device = "gloo (fallback)", torch.device("cpu")
                self.note = why or "RCCL probe failed on another rank"
        ones = torch.ones(1, dtype=torch.float64, device=self.device)
        dist.all_reduce(ones)
        self.ranks_seen = int(ones.item())

    def barrier(self):
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist
        t = torch.zeros(1, dtype=torch.float64, device=self.device)
        dist.all_reduce(t)
        t.item()                                                                # the reduction has completed on this rank

    def reduce(self, frames: float, elapsed: float, checksum: int):
        return reduce_counters(frames, elapsed, checksum, device=self.device)

    def gather(self, value: float):
        """→ [value of rank 0, value of rank 1, …] on every rank."""
        if self.world == 1:
            return [float(value)]
        import torch
        import torch.distributed as dist
        mine = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        return [float(o.item()) for o in out]

    def describe(self):
        d = {"backend": self.backend, "ranks_seen": self.ranks_seen, "world_size": self.world}
        if self.note:
            d["note"] = self.note
        return d

    def close(self):
        if self.world == 1:
            return
        import torch.distributed as dist
        if dist.is_initialized():
            self.barrier()
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------
# host-side placement: a rank's launch loop and its pinned staging buffers belong on the NUMA node its GPU hangs off
# ---------------------------------------------------------------------------------------------------------------------------
def parse_cpulist(text: str):
    """'0-3,8,10-11' (sysfs cpulist) → [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(pci_bus_id: str, sysfs: str = "/sys"):
    """(numa node, its CPUs) of the PCI device 'dddd:bb:dd.f' per sysfs; (None, []) when the platform does not say (numa_node = -1: one node, or a VM)"""
    try:
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")).read().strip())
        if node < 0:
            return None, []
        return node, parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")).read())
    except (OSError, ValueError):
        return None, []


def bind_to_gpu_numa(local_rank, sysfs: str = "/sys", pci_bus_id: str | None = None, dry_run: bool = False):
    """Restrict this process to the CPUs of the NUMA node of GPU `local_rank` (∩ its current affinity mask) — one rank per GPU on a two-socket node otherwise
    runs half of its launch loops and pinned copies across the socket interconnect.  Returns a small record for the bench detail file; never raises:
    a platform that does not expose the topology leaves the affinity alone.  local_rank=None / dry_run: only report."""
    rec = {"bound": False}
    try:
        if pci_bus_id is None and local_rank is not None:
            import torch
            p = torch.cuda.get_device_properties(local_rank)
            pci_bus_id = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        if pci_bus_id is None:
            rec["why"] = "no device"
            return rec
        rec["pci"] = pci_bus_id
        node, cpus = gpu_numa_cpus(pci_bus_id, sysfs)
        if node is None or not cpus:
            rec["why"] = "sysfs names no NUMA node for this device"
            return rec
        rec["node"] = node
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            rec["why"] = "node %d has no CPU inside this process's affinity mask" % node
            return rec
        rec["cpus"] = len(allowed)
        if not dry_run:
            os.sched_setaffinity(0, allowed)
            rec["bound"] = True
    except Exception as e:  # noqa: BLE001 — placement is an optimisation, never a failure
        rec["why"] = "%s: %s" % (type(e).__name__, e)
    return rec
