#!/usr/bin/env python3
"""bench.py — composited frames/s of the backscrub hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU, RCCL);
launched that way by someone else it just reads RANK / LOCAL_RANK / WORLD_SIZE.

A "step" is one pass of the whole per-frame hot path over one batch of device-resident synthetic camera frames: ROI
resize + BGR2RGB + bilateral + normalise, the segmentation network, decode + temporal IIR, mask upscale + 5x5 blur, alpha
blend with the background (`bsx_step_batch`).  `value` = BASELINE.json configs[1]: batch of 256 640x480 frames,
segm_lite_v681 (Google Meet 160x96).  Streams are independent, so N GPUs run N such batches (weak scaling, no data-path
collective); the only RCCL traffic is the all-reduce of the throughput counters.

Rank 0 prints ONE JSON line (contract in the task statement) that additionally carries
  roofline / roofline_blend   dominant kernel and the blend kernel, hipEvent-timed per launch (bsx_profile_batch)
  cpu_baseline                the CPU oracle port timed on this box's host cores (+ parity_sample: mask IoU vs the oracle)
  configs                     (N = 1 only) the other single-GPU BASELINE configurations, measured the same way:
                              configs[2] 256 x 1280x720 mlkit, configs[3] 1024 x 640x480 deeplab with a per-step H2D upload +
                              GPU resize of an animated background frame, configs[4]'s per-GPU slice 1024 x 1280x720 segm_full
  single_stream               latency of the drop-in path (bsx_process_host = what bs_maskgen_process forwards to)
The oracle is used only in the cpu_baseline / parity legs — never in the measured path.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP32_PEAK_TFLOPS = 157.3   # f32 vector/matrix peak
F16_PEAK_TFLOPS = 2500.0   # dense f16 MFMA peak
METRIC = "composited frames/sec at 640×480 (batch), 1/2/4/8 MI355X + mask IoU vs CPU ref"
NAMES = {"lite": "segm_lite_v681.tflite", "full": "segm_full_v679.tflite",
         "mlkit": "selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite", "deeplab": "deeplabv3_257_mv_gpu.tflite"}
PMC_NAMES = {"frame_program": ("bsx_mid", "frame_program_k"), "blend": "blend16_k", "blend(standalone)": "blend16_k", "mask_blend": "mask_tile_k<true>",
             "mask_upscale_blur": "mask_tile_k<false>", "prep": "prep_fused_k", "prep_resize": "prep_resize_k", "prep_bilateral": "prep_bilateral_k", "decode_iir": "decode_k",
             "seg_head": "seg_head_k", "seg_k2": "seg_k2_k", "seg_k3": "seg_k3_k", "seg_tail": "seg_tail_k", "seg_tail+decode": "seg_tail_k", "seg_gate": "seg_gate_k"}


def cpu_description():
    """CPU model string + socket / core counts of this box ("core count and CPU model stated")."""
    model, sockets, cores = "unknown", set(), os.cpu_count() or 1
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                sockets.add(line.split(":", 1)[1].strip())
    except OSError:
        pass
    return {"model": model, "sockets": max(len(sockets), 1), "logical_cpus": cores}


def csrc_digest():
    """Content hash of the kernel sources (backscrub_amd/csrc): what ties the committed PMC passes to the kernels that are timed.  A content hash, not a git
    tree id — the GPU box receives a snapshot without .git.  tools/merge_pmc.py stamps profiles/pmc_latest.json with the same function."""
    import hashlib
    d = os.path.join(ROOT, "backscrub_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".cpp", ".hpp")):
            h.update(fn.encode() + b"\0")
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def pipe_of(model_name, launch_name):
    """(pipe label, peak TFLOP/s of USEFUL flops) of the arithmetic pipe a launch issues on — the denominator a flops fraction may be quoted against.
    DeepLab's 1x1 convolutions (GEMM launches `conv#N`, and the expand half of the fused `conv#N+dw#M` launches) run on v_mfma_f32_16x16x32_f16 with the
    3-term split product (f32-grade): the f16 pipe does 3 MFMAs per useful MAC → 2500 / 3; BSX_F16_GEMM=fast/fast16: 1 term; =off: the f32 pipe.
    Meet / MLKit kernels use v_mfma_f32_16x16x4_f32 (157.3 TFLOP/s); depthwise / resize / argmax work is f32 VALU (157.3)."""
    if "deeplab" in model_name and launch_name.startswith("conv#") and not launch_name.startswith("conv#0+"):     # conv#0+dw#1+conv#2 = dl_head0_k: 3x3 stem on the f32 MFMA
        mode = os.environ.get("BSX_F16_GEMM", "")
        if mode == "off":
            return "f32 MFMA (v_mfma_f32_16x16x4_f32)", FP32_PEAK_TFLOPS
        if mode in ("fast", "fast16"):
            return "f16 MFMA (v_mfma_f32_16x16x32_f16), 1 term", F16_PEAK_TFLOPS
        return "f16 MFMA (v_mfma_f32_16x16x32_f16), 3-term split product: useful peak = dense f16 peak / 3", round(F16_PEAK_TFLOPS / 3.0, 1)
    return "f32 MFMA / VALU", FP32_PEAK_TFLOPS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-seconds", type=float, default=2.0, help="untimed clock ramp before the warmup steps: the step is repeated for this long so that "
                    "the GPU has left its idle power state (a step is ~0.6 ms: W = 20 of them do not; a cold box measured 380 k instead of 455 k frames/s)")
    ap.add_argument("--batch", type=int, default=256, help="streams per GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--model", default="lite", help="lite|full|mlkit|deeplab or a .tflite path")
    ap.add_argument("--per-stream-bg", action="store_true", help="every stream composites over its own background frame instead of one shared image")
    ap.add_argument("--bg-ring", action="store_true", help="animated background: every step uploads the next frame of a pinned 36-frame 480x360 ring (H2D) and resizes it on the GPU (grab_background), inside the timed region")
    ap.add_argument("--host-io", action="store_true", help="measure the with-H2D/D2H variant (per-step upload of the frames and download of the composites through pinned host buffers) over steps/2 "
                    "steps instead of the default line's 4; reported as host_io, never as value")
    ap.add_argument("--no-host-io", action="store_true", help="skip the host_io leg")
    ap.add_argument("--second-device-check", action="store_true", help="(internal, run by `--gpus N` as a subprocess of rank 0) one context on a device other than 0, checked against device 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the `configs` / `single_stream` legs (N = 1 only)")
    ap.add_argument("--no-side-probes", action="store_true", help="skip the composite_only / pipelined probes that follow the timed region (profiling runs: their launches would mix into the per-kernel averages)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--profile-iters", type=int, default=5)
    ap.add_argument("--dump-launches", default="", help="write the per-launch hipEvent table to this file")
    ap.add_argument("--selftest-dist", action="store_true", help="CPU plumbing test of the multi-process path (gloo): launch, rendezvous, counter all-reduce, JSON line — no GPU work")
    a = ap.parse_args()
    if a.no_side_probes:
        os.environ["BSX_BENCH_NO_SIDE_PROBES"] = "1"
    return a


def resolve_model(key):
    if key in NAMES:
        real = os.path.join(ROOT, "models", NAMES[key])    # model DATA the user supplies; tools/stage_models.py copies the reference's files here
        if os.path.exists(real):
            return real, NAMES[key], "reference weights"
        from tools import make_synthetic_model
        return make_synthetic_model.ensure(key), NAMES[key], "random-init weights, reference architecture"
    return key, os.path.basename(key), "user model"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """--gpus N without a launcher: become `torch.distributed.run` with N ranks on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


# ------------------------------------------------------------------------------------------------------------------------------
# CPU legs (test infrastructure: the oracle is the checker / the baseline, never the measured path)
# ------------------------------------------------------------------------------------------------------------------------------
def parity_sample(model_path, width, height, frames, bg, gpu_masks, gpu_out, need_person):
    """The metric's "mask IoU vs CPU ref" on a small sample: the first streams of the measured job (constant frames, so both
    sides are in the IIR steady state) against the CPU oracle."""
    import numpy as np
    from oracle import oracle_py
    ious, max_abs, differing, fg = [], 0, 0, []
    k = len(frames)
    for i in range(k):
        ctx = oracle_py.Ctx(model_path, width, height)
        for _ in range(4):                                   # 3 frames flush the IIR, the 4th is the steady state
            want = ctx.process(frames[i])
        ctx.close()
        fa, fb = gpu_masks[i] < 128, want < 128
        fg.append(float(fb.mean()))
        union = np.logical_or(fa, fb).sum()
        ious.append(1.0 if union == 0 else float(np.logical_and(fa, fb).sum() / union))
        comp = oracle_py.alpha_blend(bg, frames[i], want)
        d = np.abs(comp.astype(np.int16) - gpu_out[i].astype(np.int16))
        max_abs = max(max_abs, int(d.max()))
        differing += int((d > 1).any(axis=-1).sum())
    out = {"streams": k, "mask_iou_min": round(min(ious), 6), "composite_max_abs_diff": max_abs,
           "composite_pixels_off_by_more_than_1": differing, "pixels": k * width * height,
           "oracle_person_fraction": [round(v, 4) for v in fg]}
    if need_person and max(fg) < 0.05:
        out["warning"] = "oracle masks contain no person: IoU is vacuous"
    return out


def usable_cpus():
    """How many CPUs this process can really run on: the scheduler affinity mask AND the cgroup CPU quota (cpu.max / cfs_quota_us) — a container on a
    256-thread host is often given a fraction of it, while os.cpu_count() still says 256 (round 3's all-core leg: 7.6x of one thread on "256 threads")."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline(model_path, width, height, target_s):
    """Time the CPU oracle port on a bounded sample: a thread sweep 1, 2 (the reference's default `threads`, app/deepseg.cc:362), then doubling up to the CPUs
    this process can use (usable_cpus: affinity and cgroup quota, not os.cpu_count()); `value` = the best leg, `cores` = its thread count.  The port parallelises
    with OpenMP ACROSS streams (one stream — one context, created and first-touched by the thread that runs it — per thread); the reference's threads are TFLite
    intra-op threads of ONE stream — a scalar port has no intra-op parallelism, so the t-thread leg is the throughput of t cores running t streams."""
    from backscrub_amd import synth
    from oracle import oracle_py
    logical = os.cpu_count() or 1
    affinity, quota = usable_cpus()
    top = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    bg = synth.background(width, height)

    def leg(threads, budget_s):
        frames = synth.frames(threads, width, height, distinct=min(threads, 4))
        oracle_py.baseline_run(model_path, frames, bg, 1, threads)                 # warm-up (page faults, thread pool)
        sec, _, _ = oracle_py.baseline_run(model_path, frames, bg, 2, threads)     # calibration
        iters = int(max(2, min(200, budget_s / max(sec / 2, 1e-3))))
        sec, stages, _ = oracle_py.baseline_run(model_path, frames, bg, iters, threads)
        return threads * iters / sec, iters, sec, stages

    counts = sorted({1, min(2, top), top} | {t for t in (4, 8, 16, 32, 64, 128, 256) if t < top})
    if affinity > top:
        counts.append(affinity)              # one leg beyond the quota, to show that it is the quota (not the port) that caps the scaling
    legs, best = [], None
    share = target_s / (len(counts) + 1.0)
    for t in counts:
        fps, iters, sec, stages = leg(t, max(1.0, share * (2.0 if t == top else 1.0)))
        legs.append({"threads": t, "value": round(fps, 2), "unit": "frames/s", "fps_per_thread": round(fps / t, 2),
                     "sample": "%d stream(s) x %d frames, %.1f s" % (t, iters, sec)})
        if best is None or fps > best[0]:
            best = (fps, t, iters, sec, stages)
    fps, t, iters, sec, stages = best
    tot = sum(stages) or 1.0
    return {"value": round(fps, 2), "unit": "frames/s", "cores": t, "kind": "port",
            "sample": "%d streams x %d frames of %dx%d through oracle/libbs_oracle_fast.so (-O3 -mavx2 -mfma, OpenMP over streams, one context per thread), %.1f s; "
                      "best of the thread sweep in `legs`" % (t, iters, width, height, sec),
            "host": {**cpu_description(), "affinity_cpus": affinity, "cgroup_cpu_quota": quota, "logical_cpus": logical},
            "legs": legs,
            "stage_share": {k: round(v / tot, 3) for k, v in zip(("prep", "infer", "mask", "blend"), stages)}}


# ------------------------------------------------------------------------------------------------------------------------------
# one measured configuration
# ------------------------------------------------------------------------------------------------------------------------------
def load_pmc(B, W, H, model_name):
    """HBM-traffic counters of this workload from the COMMITTED rocprofv3 passes (profiles/pmc_latest.json: one entry per workload,
    tools/profile_config.sh) — not measured in this run; the line says so in `traffic_source`."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        for e in pj.get("workloads", [pj]):
            wl = e.get("workload", {})
            if (wl.get("batch"), wl.get("width"), wl.get("height"), wl.get("model")) == (B, W, H, model_name):
                e = dict(e)
                e["csrc_digest"] = pj.get("csrc_digest")          # the kernel sources the passes were collected on (tools/merge_pmc.py)
                return e
    except Exception:
        pass
    return {}


def traffic_of(pmc, launch_index, n_launches, name):
    """(2 * FETCH_SIZE + WRITE_SIZE) KiB of one launch: by position in the step when the profiled step has the same number of launches
    (layers that share a kernel instantiation — DeepLab's GEMMs — keep their own figure), else by kernel name."""
    seq = []
    for e in pmc.get("step_launches") or []:              # the fused mask + blend launch is two dispatches when the ROI does not cover the frame
        if seq and seq[-1]["kernel"].startswith("outside_roi"):
            seq[-1] = {"kernel": seq[-1]["kernel"] + " + " + e["kernel"], "FETCH_SIZE_KiB": seq[-1]["FETCH_SIZE_KiB"] + e["FETCH_SIZE_KiB"],
                       "WRITE_SIZE_KiB": seq[-1]["WRITE_SIZE_KiB"] + e["WRITE_SIZE_KiB"]}
        else:
            seq.append(dict(e))
    k = None
    if len(seq) == n_launches and 0 <= launch_index < n_launches:
        k = seq[launch_index]
    else:
        wants = PMC_NAMES.get(name, ())
        wants = (wants,) if isinstance(wants, str) else wants          # the specialised (hipRTC) and the interpreted program are different kernels
        kern = pmc.get("kernels", {})
        k = next((v for want in wants for n_, v in kern.items() if n_ == want or n_.startswith(want)), None)
    if k and "FETCH_SIZE_KiB" in k and "WRITE_SIZE_KiB" in k:
        return int((2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024), k.get("kernel")
    return None, None


def roofline_of(s, pmc, model_name, launch_index=-1, n_launches=0):
    """achieved = ALGORITHMIC bytes (or flops) of the launch / its mean hipEvent duration; traffic = HBM bytes per launch from
    the committed rocprofv3 PMC passes: (2*FETCH_SIZE + WRITE_SIZE) KiB — FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 counts
    128-B reads at 64 B).  The wall is chosen against the pipe the kernel ISSUES on (pipe_of): a launch is "mfma"-bound only if its arithmetic
    intensity exceeds that pipe's ridge; below it the line is HBM bytes.  `frac_counted_traffic` = the counted HBM bytes over the same duration —
    what really crosses HBM (algorithmic bytes that hit L2, e.g. a shared background image, are not in it)."""
    traffic, kern = traffic_of(pmc, launch_index, n_launches, s["name"]) if pmc else (None, None)
    src = {}
    if traffic is not None:
        stale = pmc.get("csrc_digest") != csrc_digest()
        src = {"traffic_source": "committed rocprofv3 --pmc passes (profiles/pmc_latest.json, round %s), not measured in this run" % pmc.get("round", "?"),
               "traffic_stale": bool(stale)}
        if stale:
            src["traffic_stale_note"] = "backscrub_amd/csrc changed since the passes were collected (digest %s then, %s now)" % (pmc.get("csrc_digest"), csrc_digest())
    if kern:
        src["traffic_kernel"] = kern
    pipe, peak_tf = pipe_of(model_name, s["name"])
    counted = {"frac_counted_traffic": round(traffic / (s["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} if traffic and s["avg_ms"] > 0 else {}
    if s["flops"] > 0 and s["flops"] / max(s["bytes"], 1) > peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9):
        a = s["flops"] / (s["avg_ms"] * 1e-3) / 1e12
        return {"kernel": s["name"], "bound": "mfma", "pipe": pipe, "achieved": round(a, 3), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(a / peak_tf, 4),
                "traffic": traffic, **src, **counted, "avg_ms": round(s["avg_ms"], 4), "algorithmic_flops_per_launch": int(s["flops"]),
                "algorithmic_bytes_per_launch": int(s["bytes"])}
    out = {"kernel": s["name"], "bound": "hbm", "achieved": round(s["GBps"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(s["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, **src, **counted, "avg_ms": round(s["avg_ms"], 4),
           "algorithmic_bytes_per_launch": int(s["bytes"])}
    if s.get("shared_bytes"):        # reads of the one background image every stream shares: algorithmic bytes, but L2 hits by construction
        hs = (s["bytes"] - s["shared_bytes"]) / (s["avg_ms"] * 1e-3) / 1e9
        out.update({"shared_background_bytes_per_launch": int(s["shared_bytes"]), "achieved_hbm_side": round(hs, 1), "frac_hbm_side": round(hs / HBM_PEAK_GBS, 4),
                    "hbm_side_note": "achieved counts every algorithmic byte, incl. the reads of the ONE background image all streams share (cache hits); *_hbm_side leaves "
                                     "them out = the bytes that must cross HBM; roofline_blend_per_stream_bg is the same kernel with every byte from HBM"})
        if out["frac"] > 1.0:        # more algorithmic bytes per second than HBM can deliver: the line is only meaningful on its HBM side
            out.update({"achieved_incl_shared_background": out["achieved"], "achieved": round(hs, 1), "frac": round(hs / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch_incl_shared_background": out["algorithmic_bytes_per_launch"],
                        "algorithmic_bytes_per_launch": int(s["bytes"] - s["shared_bytes"])})
    if "bytes_dense" in s:           # the data-dependent fused mask + blend (measure()): what the launch had to move for this input, next to SURVEY §8(d)'s dense figure
        out.update({"tiles": s["tiles"], "algorithmic_bytes_per_launch_dense_10Bpx": int(s["bytes_dense"]),
                    "achieved_dense_10Bpx": round(s["bytes_dense"] / (s["avg_ms"] * 1e-3) / 1e9, 1),
                    "note": "algorithmic bytes for THIS input: 11 B per ROI pixel on general tiles, 7 B on tiles whose mask is uniformly 0 / 255 (one operand is not read, "
                            "the mask phases are skipped), 6 B outside the ROI; the *_dense_10Bpx fields price every pixel at SURVEY 8(d)'s 10 B and are NOT a bandwidth"})
    if s["flops"] > 0:
        a = s["flops"] / (s["avg_ms"] * 1e-3) / 1e12
        out.update({"algorithmic_flops_per_launch": int(s["flops"]), "flops_pipe": pipe, "flops_frac_of_pipe": round(a / peak_tf, 4),
                    "intensity_flop_per_byte": round(s["flops"] / max(s["bytes"], 1), 1), "ridge_flop_per_byte": round(peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9), 1)})
    return out


def finish_counters(coll, frames, elapsed, checksum):
    """The job's one reduction, shared by measure() and the CPU self-test: {frames Σ, elapsed max, checksum Σ} + every rank's own rate.  coll=None = THIS RANK ALONE
    (N = 1, or rank 0's solo reference run while the other ranks wait at a barrier): nothing collective may be touched then — an all-reduce entered by one rank
    only would pair up with the waiting ranks' barrier (round 4, found by `bench.py --gpus 2` on one GPU: it hung)."""
    if coll is None:
        return float(frames), float(elapsed), int(checksum) % (1 << 40), [frames / elapsed]
    total, max_elapsed, checksum_all = coll.reduce(frames, elapsed, checksum)
    return total, max_elapsed, checksum_all, coll.gather(frames / elapsed)


def solo_reference(coll, rank, fn):
    """rank 0 runs fn() ALONE — fn must not touch the collective (coll=None inside) — while the other ranks wait; → fn()'s value on rank 0, None elsewhere"""
    v = fn() if rank == 0 else None
    coll.barrier()
    return v


def measure(model_key, W, H, B, steps, warmup, rank, world, local_rank, per_stream_bg=False, bg_ring=False, profile_iters=5, dump_launches="", ramp_s=0.0, coll=None,
            profile=True):
    """Run one configuration on this rank's GPU.  Returns a dict with the timed result and (rank 0) the per-launch profile and
    the samples the parity leg needs.  `coll` (backscrub_amd.dist.Collective) carries the barriers around the timed region and the one
    reduction of the job; None = this rank alone (N = 1, and rank 0's solo reference run at N > 1)."""
    import numpy as np
    import torch

    import backscrub_amd
    from backscrub_amd import synth

    model_path, model_name, weights = resolve_model(model_key)
    mg = backscrub_amd.MaskGen(model_path, W, H, n_streams=B, device=local_rank)
    # synthetic, device-resident inputs: each GPU owns its own B streams (seeded by global stream id); at 640x480 the first two
    # streams carry the two REAL webcam frames of tests/golden/photo_2x640x480.png so that the parity sample has a real person
    distinct = 16
    host = synth.frames(distinct, W, H, t=rank)
    photo = False
    if (W, H) == (640, 480):
        try:
            from tools import make_photo_fixture
            host[:2] = make_photo_fixture.load_frames()
            photo = True
        except Exception:
            pass
    d_base = torch.from_numpy(host).cuda()
    d_frames = d_base.repeat((B + distinct - 1) // distinct, 1, 1, 1)[:B].contiguous()
    bg_host = synth.background(W, H, seed=1 + rank)
    d_bg = torch.from_numpy(bg_host).cuda()
    if per_stream_bg:      # [B,H,W,3]: one background frame per stream, rolled so that no two streams share bytes
        d_bg = torch.stack([torch.roll(d_bg, shifts=3 * i, dims=1) for i in range(min(B, 64))]).repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
    d_out = torch.empty_like(d_frames)
    ring = None
    if bg_ring:
        # animated background (configs[3]): the reference decodes the video on a host thread (app/background.cc:29-104) and
        # grab_background() resizes the current frame to the camera size EVERY frame (:186).  webm cannot be decoded in this image:
        # the decode is emulated by a ring of 36 pre-decoded 480x360 frames (the size of backgrounds/animated.gif) in PINNED host
        # memory; per step: H2D of the next frame + bsx_resize_bgr on the GPU, both inside the timed region.
        ring = torch.from_numpy(np.stack([synth.background(480, 360, seed=100 + i) for i in range(36)])).pin_memory()
        d_small = torch.empty((1, 360, 480, 3), dtype=torch.uint8, device="cuda")

    def one_step(t):
        if ring is not None:
            d_small[0].copy_(ring[t % 36], non_blocking=True)
            bg = mg.resize_bgr(d_small, W, H)[0]
            mg.step(d_frames, bg, d_out)
        else:
            mg.step(d_frames, d_bg, d_out)

    def barrier():
        torch.cuda.synchronize()
        if coll is not None:
            coll.barrier()

    if ramp_s > 0:                        # clock ramp (untimed, before the W warmup steps): sustained work until the GPU is out of its idle state
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < ramp_s:
            for t in range(8):
                one_step(t)
            torch.cuda.synchronize()
    for t in range(warmup):
        one_step(t)
    barrier()
    t0 = time.perf_counter()
    for t in range(steps):
        one_step(warmup + t)
    barrier()
    elapsed = time.perf_counter() - t0

    checksum = int(d_out[:, ::16, ::16].to(torch.int64).sum().item())
    total_frames, max_elapsed, checksum_all, rank_fps = finish_counters(coll, B * steps, elapsed, checksum)     # the only collective of the job
    res = {"model_path": model_path, "model_name": model_name, "weights": weights, "B": B, "W": W, "H": H, "photo": photo,
           "fps": total_frames / max_elapsed, "ms_per_step": 1e3 * max_elapsed / steps, "checksum": checksum_all, "mg": mg, "rank_fps": rank_fps,
           "d_frames": d_frames, "d_bg": d_bg, "d_out": d_out, "host": host, "bg_host": bg_host}
    if rank == 0 and profile:
        k = 4
        if ring is not None:                 # parity needs ONE known background: re-run the last step over the still image
            mg.step(d_frames, d_bg, d_out)
            torch.cuda.synchronize()
        res["masks_k"] = mg.masks()[:k].cpu().numpy()
        res["out_k"] = d_out[:k].cpu().numpy()
        # the same step without STORING the full-resolution mask (BSX_STEP_NO_MASK: 6 instead of 7 HBM bytes per pixel in the last launch) — reported beside
        # `value`, never as it: the headline materialises the mask, as bs_maskgen_process does
        if ring is None and not per_stream_bg and not os.environ.get("BSX_BENCH_NO_SIDE_PROBES"):
            probe = max(3, min(steps, 100))
            d_probe = torch.empty_like(d_out)
            for t in range(3):
                mg.step_ex(d_frames, d_bg, d_probe, no_mask=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for t in range(probe):
                mg.step_ex(d_frames, d_bg, d_probe, no_mask=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            mg.step(d_frames, d_bg, d_out)                         # same temporal state → the composite must be the same bytes
            torch.cuda.synchronize()
            res["composite_only"] = {"what": "bsx_step_batch_ex(BSX_STEP_NO_MASK): composite written, full-resolution mask not stored", "steps": probe,
                                     "value": round(B * probe / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / probe, 4),
                                     "composite_identical_to_the_storing_step": bool(torch.equal(d_probe, d_out))}
            # the same step as a two-deep pipeline (bsx_step_batch_pipelined): call k enqueues the mask pipeline of batch k and, on the context's own stream, the
            # composite of batch k - 1 — the HBM-bound half under the latency-bound half, as the reference's CalcMask worker runs next to its blend loop
            # (app/deepseg.cc:159-285).  `probe` calls = `probe` whole steps of work (the pipeline is primed before the clock starts and still holds one batch when
            # it stops).  Reported beside `value`, never as it: `value` is the synchronous step.
            try:
                for t in range(3):
                    mg.step_pipelined(d_frames, d_bg, d_probe)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for t in range(probe):
                    mg.step_pipelined(d_frames, d_bg, d_probe)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                mg.flush_pipelined()
                mg.step(d_frames, d_bg, d_out)
                torch.cuda.synchronize()
                res["pipelined"] = {"what": "bsx_step_batch_pipelined: mask pipeline of batch k on the caller's stream || mask tiles + blend of batch k - 1 on a low-priority "
                                            "stream of the context; results bit-identical to the synchronous step, one call later", "steps": probe,
                                    "value": round(B * probe / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / probe, 4),
                                    "speedup_vs_value": round((B * probe / dt) / (total_frames / max_elapsed), 3) if world == 1 else None,
                                    "composite_identical_to_the_synchronous_step": bool(torch.equal(d_probe, d_out))}
            except Exception as e:  # noqa: BLE001 — an extra figure must not take the headline down
                res["pipelined"] = {"error": str(e)[:200]}
                try:
                    mg.flush_pipelined()
                except Exception:  # noqa: BLE001
                    pass
            del d_probe
        # EVERY stream of the batch, on the GPU: streams i and i + 16 carry the same scene (and, shared background, the same temporal history), so their
        # masks and composites must be identical bytes whatever tile / workgroup / XCD they ran on; the first streams are then held to the oracle (parity_sample)
        if per_stream_bg:
            res["full_batch"] = None
        else:
            groups_ok = 0
            n_groups = B // distinct
            masks_all = mg.masks()[:B]
            for gi in range(1, n_groups):
                if torch.equal(d_out[gi * distinct:(gi + 1) * distinct], d_out[:distinct]) and torch.equal(masks_all[gi * distinct:(gi + 1) * distinct], masks_all[:distinct]):
                    groups_ok += 1
            res["full_batch"] = {"streams": B, "distinct_scenes": distinct, "groups_compared_with_group_0": max(n_groups - 1, 0), "groups_identical": groups_ok,
                                 "all_identical": groups_ok == max(n_groups - 1, 0)}
        stats = mg.profile(d_frames, d_bg, d_out, iters=profile_iters)
        # The fused mask + blend launch is data dependent since round 4: a tile whose whole model-resolution source block is 0xFF / 0x00 (the temporal filter's steady
        # state away from the person's outline) skips the mask phases and reads only the operand its composite is a copy of.  Its algorithmic bytes are therefore
        # stated for THIS input: per ROI pixel 11 B on a general tile (background 3 + frame 3 read, composite 3 + mask 1 written), 7 B on a uniform one; 6 B outside
        # the ROI (background copied).  `bytes_dense` keeps SURVEY §8(d)'s 10 B/px figure.
        try:
            ts = mg.mask_tile_stats(B)
            i_ = mg.info
            roi_px, in_roi_px = i_["roi"][2] * i_["roi"][3], i_["in_roi"][2] * i_["in_roi"][3]
            f_uni = (ts["uniform_255"] + ts["uniform_0"]) / max(ts["tiles"], 1)
            aware = B * (in_roi_px + 6.0 * (W * H - roi_px) + roi_px * (11.0 * (1.0 - f_uni) + 7.0 * f_uni))
            res["mask_tiles"] = {k: ts[k] for k in ("tiles", "uniform_255", "uniform_0", "general", "tile")}
            for s in stats:
                if s["name"] == "mask_blend":
                    s["bytes_dense"] = s["bytes"]
                    s["bytes"] = aware
                    s["tiles"] = {k: ts[k] for k in ("tiles", "uniform_255", "uniform_0", "general", "tile")}
        except Exception:  # noqa: BLE001 — accounting only
            pass
        # ONE background image shared by all B streams (0.9 MB at VGA) is cache-resident by construction: its reads are part of the algorithmic bytes (10 B/px)
        # but cannot be HBM traffic.  Stated per launch so that every blend line also carries its HBM-SIDE figure (roofline_of: *_hbm_side).
        if not per_stream_bg:
            try:
                i_ = mg.info
                roi_px = i_["roi"][2] * i_["roi"][3]
                ts_ = res.get("mask_tiles")
                f_bg = (ts_["general"] + ts_["uniform_255"]) / max(ts_["tiles"], 1) if ts_ else 1.0      # tiles that read the background at all
                for s in stats:
                    if s["name"] == "mask_blend":
                        s["shared_bytes"] = B * 3.0 * ((W * H - roi_px) + roi_px * f_bg)
                    elif s["name"].startswith("blend"):
                        s["shared_bytes"] = B * 3.0 * W * H
            except Exception:  # noqa: BLE001 — accounting only
                pass
        for s in stats:
            s["GBps"] = s["bytes"] / (s["avg_ms"] * 1e-3) / 1e9 if s["avg_ms"] > 0 else 0.0
        extra = [s for s in stats if s["name"].endswith("(standalone)")]   # measured for its roofline line, not part of the step
        stats = [s for s in stats if not s["name"].endswith("(standalone)") and "(inside the launch before)" not in s["name"]]   # fused-away steps launch nothing
        if dump_launches:
            with open(dump_launches, "w") as f:
                f.write(mg.plan())
                for i, s in enumerate(stats):
                    f.write("%3d %-22s %8.2f us %9.1f GB/s %8.2f GFLOP/s\n" % (i, s["name"], s["avg_ms"] * 1e3, s["GBps"], s["flops"] / max(s["avg_ms"], 1e-9) / 1e6))
        groups = {"prep": 0.0, "network": 0.0, "decode": 0.0, "mask": 0.0, "blend": 0.0}
        for s in stats:
            g = {"prep_resize": "prep", "prep_bilateral": "prep", "prep": "prep", "decode_iir": "decode", "mask_upscale_blur": "mask", "blend": "blend",
                 "mask_blend": "blend"}.get(s["name"], "network")
            groups[g] += s["avg_ms"]
        net = [s for s in stats if {"prep_resize": 1, "prep_bilateral": 1, "prep": 1, "decode_iir": 1, "mask_upscale_blur": 1, "blend": 1, "mask_blend": 1}.get(s["name"]) is None]
        res.update(stats=stats, extra=extra, groups=groups, net=net, net_ms=sum(s["avg_ms"] for s in net), net_flops=sum(s["flops"] for s in net),
                   net_launches=len(net))
    return res


def summarize(res, pmc):
    """→ the JSON fragment of one measured configuration (rank 0)."""
    stats, extra = res["stats"], res["extra"]
    dom = max(stats, key=lambda s: s["avg_ms"])
    blend = dict((extra or [s for s in stats if s["name"] in ("blend", "mask_blend")])[0], name="blend")
    n_l = len(stats)
    out = {"value": round(res["fps"], 1), "unit": "frames/s", "ms_per_step": round(res["ms_per_step"], 4),
           "roofline": roofline_of(dom, pmc, res["model_name"], stats.index(dom), n_l), "roofline_blend": roofline_of(blend, pmc, res["model_name"])}
    if extra:
        out["roofline_blend"]["note"] = "bsx_composite_batch kernel timed stand-alone; inside the step the blend is fused with mask upscale+blur (mask_blend)"
    if res["net_launches"] > 1:              # the network launches together, against BOTH walls: useful flops over the pipe they issue on, and bytes over HBM
        ms = res["net_ms"] * 1e-3
        net_traffic = [traffic_of(pmc, i, n_l, s_["name"])[0] for i, s_ in enumerate(stats) if s_ in res["net"]] if pmc and len(pmc.get("step_launches") or []) else []
        traffic = int(sum(net_traffic)) if net_traffic and all(t is not None for t in net_traffic) else None
        # pipe time: every launch's flops at the peak of ITS pipe (DeepLab: GEMM launches on the split-f16 pipe, the rest f32)
        pipe_s = sum(s_["flops"] / (pipe_of(res["model_name"], s_["name"])[1] * 1e12) for s_ in res["net"])
        alg_bytes = sum(s_["bytes"] for s_ in res["net"])
        f_pipe = pipe_s / ms
        f_alg = alg_bytes / ms / 1e9 / HBM_PEAK_GBS
        f_cnt = traffic / ms / 1e9 / HBM_PEAK_GBS if traffic else None
        hbm_side = max(f_alg, f_cnt or 0.0)
        out["roofline_network"] = {"kernel": "network (%d launches)" % res["net_launches"], "bound": "hbm" if hbm_side >= f_pipe else "mfma",
                                   "frac": round(max(hbm_side, f_pipe), 4),
                                   "flops_TFLOPs": round(res["net_flops"] / ms / 1e12, 3), "frac_of_issuing_pipes": round(f_pipe, 4),
                                   "algorithmic_GBps": round(alg_bytes / ms / 1e9, 1), "frac_hbm_algorithmic": round(f_alg, 4),
                                   "counted_GBps": round(traffic / ms / 1e9, 1) if traffic else None, "frac_hbm_counted_traffic": round(f_cnt, 4) if f_cnt else None,
                                   "peak_hbm_GBps": HBM_PEAK_GBS, "traffic": traffic, "avg_ms": round(res["net_ms"], 4),
                                   "note": "frac_of_issuing_pipes = sum over launches of useful flops / the peak of the pipe that launch issues on (pipe_of), over the measured time"}
    if res.get("mask_tiles") is not None:      # how the fused mask + blend launch classified its tiles for this input (uniform tiles skip the mask phases and one operand)
        out["mask_tiles"] = res["mask_tiles"]
    if res.get("composite_only") is not None:
        out["composite_only"] = res["composite_only"]
    if res.get("pipelined") is not None:
        out["pipelined"] = res["pipelined"]
    if res.get("full_batch") is not None:
        out["full_batch_twin_streams"] = res["full_batch"]
    out["stage_ms"] = {k: round(v, 4) for k, v in res["groups"].items()}
    out["stage_ms"]["sum_of_launches"] = round(sum(s["avg_ms"] for s in stats), 4)
    out["top_launches"] = [{"name": s["name"], "ms": round(s["avg_ms"], 4), "GBps": round(s["GBps"], 1)} for s in sorted(stats, key=lambda s: -s["avg_ms"])[:8]]
    return out


def release(res):
    import torch
    res["mg"].close()
    for k in ("mg", "d_frames", "d_bg", "d_out"):
        res.pop(k, None)
    torch.cuda.empty_cache()


def single_stream_latency(model_key, W, H, calls):
    """The drop-in path an unchanged deepseg.cc would take: bs_maskgen_process → bsx_process_host (H2D of one frame, the whole
    mask pipeline for one stream, D2H of the mask, synchronous).  Context: the reference's README quotes ~10 FPS for DeepLab on two
    CPU cores (README.md:177) and the Meet model card ~120 FPS inference on a laptop CPU — neither measured here."""
    import numpy as np

    import backscrub_amd
    from backscrub_amd import synth
    path, name, _ = resolve_model(model_key)
    mg = backscrub_amd.MaskGen(path, W, H, n_streams=1)
    f = synth.frame(W, H, 0)
    mask = np.empty((H, W), np.uint8)
    for _ in range(5):
        mg.process_host(f, 0, mask)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        mg.process_host(f, 0, mask)
        ts.append(1e3 * (time.perf_counter() - t0))
    mg.close()
    ts.sort()
    return {"network": name, "frame": "%dx%d" % (W, H), "calls": calls, "p50_ms": round(ts[len(ts) // 2], 3), "p99_ms": round(ts[min(len(ts) - 1, int(len(ts) * 0.99))], 3),
            "fps_at_p50": round(1e3 / ts[len(ts) // 2], 1)}


def multi_gpu_sections(coll, main, c4, solo1, solo4, world):
    """The N > 1 part of the line (rank 0): what RCCL really connected, the per-rank rates behind `value`, the north-star job's per-GPU slice
    (BASELINE configs[4]: 8192 x 1280x720 segm_full streams over 8 GPUs = 1024 per GPU, weak-scaled to N) and both against rank 0 running the
    same work ALONE on this box a moment earlier (the other ranks waiting at a barrier) — the driver computes its own efficiency from separate runs;
    this one is same-box, same-minute.  `main` / `c4`: {"fps", "ms_per_step", "rank_fps"}; solo*: frames/s or None."""
    def leg(r, solo, what):
        fps = r["rank_fps"]
        d = {"workload": what, "value": round(r["fps"], 1), "unit": "frames/s", "ms_per_step": round(r["ms_per_step"], 4),
             "per_rank_fps": [round(v, 1) for v in fps], "per_rank_fps_min": round(min(fps), 1), "per_rank_fps_max": round(max(fps), 1)}
        if solo:
            d["rank0_alone_fps"] = round(solo, 1)
            d["efficiency_vs_rank0_alone"] = round(r["fps"] / (world * solo), 4)
        return d
    out = {"collective": coll.describe(), "ranks_seen": coll.ranks_seen,
           "configs1": leg(main, solo1, "BASELINE configs[1] per GPU: %d x %dx%d, %s" % (main["B"], main["W"], main["H"], main["model_name"]))}
    if c4 is not None:
        out["configs4"] = leg(c4, solo4, "BASELINE configs[4]: %d x %dx%d %s streams sharded as contiguous blocks of %d per GPU over %d GPU(s), no data-path collective"
                              % (c4["B"] * world, c4["W"], c4["H"], c4["model_name"], c4["B"], world))
        out["configs4"]["streams_total"] = c4["B"] * world
    return out


def selftest_dist(args):
    """CPU plumbing check of the multi-process path (tests/test_dist_gloo.py): the same launch, rendezvous, Collective (gloo instead of RCCL),
    barriers, counter reduction, per-rank gather and JSON assembly (multi_gpu_sections) as the GPU run — the GPU work replaced by made-up timings."""
    from backscrub_amd.dist import Collective, shard_streams
    coll = Collective(gpu=False)
    world, rank = coll.world, coll.rank

    def fake(B, W, H, name, steps, ms, c):                # what measure() does around its timed region, with a rank-dependent made-up duration
        if c is not None:
            c.barrier()
        elapsed = steps * ms * 1e-3 * (1.0 + 0.25 * rank)
        if c is not None:
            c.barrier()
        frames, max_elapsed, checksum, rank_fps = finish_counters(c, B * steps, elapsed, 1000 + rank)
        return {"B": B, "W": W, "H": H, "model_name": name, "fps": frames / max_elapsed, "ms_per_step": 1e3 * max_elapsed / steps,
                "rank_fps": rank_fps, "frames": frames, "elapsed_max": max_elapsed, "checksum": checksum}
    a, b = shard_streams(args.batch * world, world, rank)
    n4 = max(3, args.steps // 4)
    # the same order of solo runs, barriers and collectives as main()
    solo1 = solo_reference(coll, rank, lambda: fake(b - a, args.width, args.height, NAMES["lite"], args.steps, 1.0, None)["fps"]) if world > 1 else None
    main = fake(b - a, args.width, args.height, NAMES["lite"], args.steps, 1.0, coll)
    solo4 = solo_reference(coll, rank, lambda: fake(1024, 1280, 720, NAMES["full"], n4, 4.0, None)["fps"]) if world > 1 else None
    c4 = fake(1024, 1280, 720, NAMES["full"], n4, 4.0, coll)
    line = None
    if rank == 0:
        line = {"metric": METRIC, "selftest": "dist", "value": main["fps"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": main["ms_per_step"], "higher_is_better": True, "scaling": "weak", "frames": main["frames"], "elapsed_max": main["elapsed_max"],
                "checksum": main["checksum"], "host": cpu_description()}
        line.update(multi_gpu_sections(coll, main, c4, solo1, solo4, world))
    coll.close()
    if rank == 0:
        if not args.no_cpu_baseline:                      # kept at world > 1 (rank 0, after the other ranks have left)
            try:
                path, _, _ = resolve_model(args.model)
                line["cpu_baseline"] = cpu_baseline(path, args.width, args.height, args.cpu_seconds)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)


def second_device_check():
    """`bench.py --second-device-check` (a subprocess of rank 0 at N > 1, so that a fault cannot take the measurement down): the code path with device != 0 —
    per-device kernel attributes, hipRTC module load, hipGraph capture of the host path, every entry point's device guard — executed on real hardware the first
    time two devices are visible.  One context on the LAST visible device and one on device 0, same frames: masks, composites and the single-frame host path
    must be byte-identical, and the caller's current device must be untouched."""
    import numpy as np
    import torch

    import backscrub_amd
    from backscrub_amd import synth
    ndev = backscrub_amd.lib().bsx_device_count()
    if ndev < 2:
        print(json.dumps({"ran": False, "why": "%d visible device(s)" % ndev}))
        return
    out = {"ran": True, "devices": [0, ndev - 1], "models": {}}
    ok_all = True
    W, H, n = 640, 480, 4
    frames = np.stack([synth.frame(W, H, s, 0) for s in range(n)])
    bg = synth.background(W, H)
    torch.cuda.set_device(0)
    for key in ("lite", "deeplab"):
        path, name, _ = resolve_model(key)
        got = {}
        for dev in (0, ndev - 1):
            mg = backscrub_amd.MaskGen(path, W, H, n_streams=n, device=dev)
            with torch.cuda.device(dev):
                d = torch.from_numpy(frames).cuda()
                d_bg = torch.from_numpy(bg).cuda()
                o = torch.empty_like(d)
                for _ in range(3):
                    mg.step(d, d_bg, o)
                torch.cuda.synchronize()
                masks, comp = mg.masks().cpu().numpy(), o.cpu().numpy()
            restored = torch.cuda.current_device() == 0
            hm = np.empty((H, W), np.uint8)
            mg.reset()
            torch.cuda.synchronize(dev)
            for _ in range(3):
                mg.process_host(frames[1], 1, hm)          # the drop-in path (hipGraph capture + replay) on that device
            got[dev] = (masks, comp, hm.copy(), restored and torch.cuda.current_device() == 0)
            mg.close()
        a_, b_ = got[0], got[ndev - 1]
        res = {"masks_identical": bool(np.array_equal(a_[0], b_[0])), "composites_identical": bool(np.array_equal(a_[1], b_[1])),
               "host_path_identical": bool(np.array_equal(a_[2], b_[2])), "host_path_equals_batch_path": bool(np.array_equal(b_[2], b_[0][1])),
               "callers_device_restored": bool(a_[3] and b_[3]), "person_fraction": round(float((b_[0] < 128).mean()), 4)}
        ok_all = ok_all and all(v for k, v in res.items() if k != "person_fraction")
        out["models"][name] = res
    out["ok"] = ok_all
    print(json.dumps(out))


def run_second_device_check():
    import subprocess
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                "TORCHELASTIC_RUN_ID", "ROLE_RANK", "ROLE_WORLD_SIZE")}
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--second-device-check"], env=env, capture_output=True, text=True, timeout=240)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"ran": True, "ok": False, "returncode": p.returncode, "stderr_tail": p.stderr[-600:]}
    except Exception as e:  # noqa: BLE001
        return {"ran": True, "ok": False, "error": repr(e)}


def main():
    args = parse()
    if args.second_device_check:
        return second_device_check()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                     # does not return
    if args.selftest_dist:
        return selftest_dist(args)
    import torch

    from backscrub_amd.dist import Collective

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("--gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists)")
    n_dev = torch.cuda.device_count()
    shared = world > n_dev                                     # more ranks than GPUs (e.g. `--gpus 2` on a 1-GPU box): ranks share devices round-robin — a plumbing run,
    local_rank = local_rank % n_dev                            # labelled `gpu_sharing` in the line; RCCL needs one GPU per rank, so the counters then travel over gloo
    torch.cuda.set_device(local_rank)
    coll = Collective(gpu=not shared) if world > 1 else None   # RCCL over xGMI, probed; labelled gloo fall-back if RCCL cannot be brought up
    if coll is not None and shared:
        coll.backend, coll.note = "gloo (ranks share GPUs)", "%d ranks on %d visible GPU(s): RCCL requires one GPU per rank" % (world, n_dev)

    W, H, B = args.width, args.height, args.batch
    default_job = (args.model, W, H, B) == ("lite", 640, 480, 256) and not args.per_stream_bg and not args.bg_ring
    solo1 = solo4 = None
    if coll is not None:
        # rank 0 ALONE first (the others wait at the barrier): the N = 1 reference of this box, minutes — not runs — apart from the N-rank number
        def solo_main():
            r1 = measure(args.model, W, H, B, args.steps, args.warmup, 0, 1, local_rank, ramp_s=args.ramp_seconds, per_stream_bg=args.per_stream_bg, bg_ring=args.bg_ring,
                         profile=False)
            fps = r1["fps"]
            release(r1)
            return fps
        solo1 = solo_reference(coll, rank, solo_main)
    res = measure(args.model, W, H, B, args.steps, args.warmup, rank, world, local_rank, ramp_s=args.ramp_seconds, per_stream_bg=args.per_stream_bg, bg_ring=args.bg_ring,
                  profile_iters=args.profile_iters, dump_launches=args.dump_launches, coll=coll)
    multi = None
    c4_frag = c4_samples = None
    if coll is not None:
        main_leg = {k: res[k] for k in ("B", "W", "H", "model_name", "fps", "ms_per_step", "rank_fps")}
        c4 = None
        if default_job and not args.no_extra_configs:
            # the north-star job's per-GPU slice on EVERY rank: BASELINE configs[4] = 8192 x HD segm_full streams / 8 GPUs = 1024 per GPU
            n4 = max(3, args.steps // 4)
            kw4 = dict(model_key="full", W=1280, H=720, B=1024)
            try:
                def solo_c4():
                    r4s = measure(steps=n4, warmup=2, rank=0, world=1, local_rank=local_rank, profile=False, **kw4)
                    fps = r4s["fps"]
                    release(r4s)
                    return fps
                solo4 = solo_reference(coll, rank, solo_c4)
                r4 = measure(steps=n4, warmup=2, rank=rank, world=world, local_rank=local_rank, profile_iters=2, coll=coll, **kw4)
                c4 = {k: r4[k] for k in ("B", "W", "H", "model_name", "fps", "ms_per_step", "rank_fps")}
                if rank == 0:
                    c4_frag = summarize(r4, load_pmc(1024, 1280, 720, r4["model_name"]))
                    c4_samples = (r4["model_path"], r4["host"][:2].copy(), r4["bg_host"], r4["masks_k"], r4["out_k"], r4["photo"])
                release(r4)
            except Exception as e:  # noqa: BLE001 — every rank takes the same path: the exception classes here are allocation / model errors, identical on all ranks
                c4 = None
                if rank == 0:
                    c4_frag = {"error": repr(e)}
        if rank == 0:
            multi = multi_gpu_sections(coll, main_leg, c4, solo1, solo4, world)
            if c4_frag is not None and "configs4" in multi:
                multi["configs4"].update({k: v for k, v in c4_frag.items() if k in ("roofline", "stage_ms", "top_launches", "full_batch_twin_streams", "error")})
        coll.close()                                          # ranks > 0 are done; rank 0 goes on alone (parity, CPU baseline, the device check)
    result = None
    if rank == 0:
        import backscrub_amd
        mode = backscrub_amd.bs_tensorflow_version()
        result = {
            "metric": METRIC, "value": round(res["fps"], 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (network) / u8 (image kernels)", "data": "synthetic (%s)" % res["weights"],
            "config": {"workload": "BASELINE configs[1]: batch=%d %dx%d frames, %s; 1 step = whole per-frame hot path (prep+network+decode+mask+blend), inputs resident in HBM; "
                                   "mask IoU vs the CPU oracle: cpu_baseline.parity_sample" % (B, W, H, res["model_name"]),
                       "streams_per_gpu": B, "frame": "%dx%d" % (W, H), "network": res["model_name"], "sharding": "streams/%d GPUs, no data-path collective" % world,
                       "library": mode},
            "checksum": res["checksum"], "host": cpu_description(),
        }
        result.update({k: v for k, v in summarize(res, load_pmc(B, W, H, res["model_name"])).items() if k not in ("value", "unit", "ms_per_step")})
        if multi is not None:
            result.update(multi)
            if shared:
                result["gpu_sharing"] = "%d ranks on %d GPU(s): throughput numbers of this line are a plumbing run, not a scaling measurement" % (world, n_dev)

    # PCIe-inclusive variant: every step uploads its frames and downloads its composites through pinned buffers.
    # Copies run on their own HIP streams with double-buffered device frames / composites, so the upload of step t+1 and the
    # download of step t-1 overlap the compute of step t (the per-stream mask state keeps the compute steps in order).
    if rank == 0 and world == 1 and not args.no_host_io:
        from backscrub_amd import synth
        mg, d_frames, d_bg, d_out = res["mg"], res["d_frames"], res["d_bg"], res["d_out"]
        h_in = torch.from_numpy(synth.frames(B, W, H, distinct=16)).pin_memory()
        h_out = torch.empty_like(h_in).pin_memory()
        s_in, s_out, s_cmp = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
        bufs = [(torch.empty_like(d_frames), torch.empty_like(d_out)) for _ in range(2)]
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_cmp = [torch.cuda.Event() for _ in range(2)]
        ev_out = [torch.cuda.Event() for _ in range(2)]

        def run(steps):
            for t in range(steps):
                fr, out = bufs[t & 1]
                with torch.cuda.stream(s_in):
                    s_in.wait_event(ev_cmp[t & 1])            # the step that last read this frame buffer has finished
                    fr.copy_(h_in, non_blocking=True)
                    ev_in[t & 1].record(s_in)
                s_cmp.wait_event(ev_in[t & 1])
                s_cmp.wait_event(ev_out[t & 1])               # the composite buffer has been downloaded
                mg.step(fr, d_bg, out)
                ev_cmp[t & 1].record(s_cmp)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_cmp[t & 1])
                    h_out.copy_(out, non_blocking=True)
                    ev_out[t & 1].record(s_out)
            torch.cuda.synchronize()

        run(2)
        io_steps = max(4, args.steps // 2) if args.host_io else 4      # the default line carries a 4-step leg (device-resident AND with-H2D/D2H variants)
        t1 = time.perf_counter()
        run(io_steps)
        dt = time.perf_counter() - t1
        result["host_io"] = {"value": round(B * io_steps / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / io_steps, 3), "steps": io_steps,
                             "note": "same step + H2D of %d frames and D2H of %d composites per step (pinned buffers, copy streams overlapped with "
                                     "compute, double-buffered); %.1f GB/s each way" % (B, B, B * W * H * 3 / (dt / io_steps) / 1e9)}
        del bufs, h_in, h_out

    # the same step with the composite leaving as YUYV 4:2:2 (convert_rgb_to_yuyv fused into the blend epilogue): reported
    # next to `value`, never as `value`; checked bit for bit against the two-call form on the last step
    if rank == 0 and world == 1 and W % 2 == 0 and not args.no_extra_configs:
        mg, d_frames, d_bg, d_out = res["mg"], res["d_frames"], res["d_bg"], res["d_out"]
        d_yuyv = torch.empty((B, H, W, 2), dtype=torch.uint8, device="cuda")
        mg.reset(); mg.step(d_frames, d_bg, d_out); a = mg.bgr_to_yuyv(d_out)          # both forms from the same (fresh) temporal state
        mg.reset(); mg.step_yuyv(d_frames, d_bg, d_yuyv)
        same = bool(torch.equal(a, d_yuyv))
        for _ in range(args.warmup):
            mg.step_yuyv(d_frames, d_bg, d_yuyv)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            mg.step_yuyv(d_frames, d_bg, d_yuyv)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        result["yuyv_out"] = {"value": round(B * args.steps / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / args.steps, 4),
                              "bit_identical_to_step_then_bgr_to_yuyv": same,
                              "note": "bsx_step_batch_yuyv: composite written as YUYV (2 B/px instead of 3), no separate packing pass"}
        del d_yuyv, a

    # `-p bgblur:25` without `-b` (deepseg.cc:652-661): background = GaussianBlur of the stream's own frame.  One pass (BSX_STEP_BGBLUR: blur tile → blend out of LDS)
    # against the two-call form (bsx_gaussian_blur_bgr into a per-stream background, then bsx_step_batch)
    if rank == 0 and world == 1 and not args.no_extra_configs and W % 4 == 0:
        mg, d_frames, d_out = res["mg"], res["d_frames"], res["d_out"]
        d_two = torch.empty_like(d_out)
        d_blur = torch.empty_like(d_frames)

        def two_call():
            mg.gaussian_blur(d_frames, 25, out=d_blur)
            mg.step(d_frames, d_blur, d_two)

        def one_pass():
            mg.step_ex(d_frames, None, d_out, bgblur=25)

        def timed(fn, iters=max(3, args.steps // 4)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t1) / iters
        ms_two, ms_one = timed(two_call), timed(one_pass)
        result["bgblur_step"] = {"ksize": 25, "ms_per_step": round(ms_one, 4), "value": round(B / (ms_one * 1e-3), 1), "unit": "frames/s",
                                 "two_call_ms_per_step": round(ms_two, 4), "speedup": round(ms_two / ms_one, 3),
                                 "note": "bsx_step_batch_ex(BSX_STEP_BGBLUR(25)): blur + blend in one pass over the frames vs bsx_gaussian_blur_bgr + bsx_step_batch"}
        del d_two, d_blur

    # the blend kernel with ONE BACKGROUND PER STREAM (animated backgrounds): nothing of its 10 B/px comes out of L2, unlike the shared
    # 0.9 MB picture of the default job whose roofline_blend line is flattered by cache hits (its PMC traffic is below the algorithmic bytes)
    if rank == 0 and world == 1 and not args.no_extra_configs and not args.per_stream_bg:
        mg, d_frames, d_out = res["mg"], res["d_frames"], res["d_out"]
        d_bg_ps = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            mg.composite(d_bg_ps, d_frames, None, d_out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            mg.composite(d_bg_ps, d_frames, None, d_out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        by = 10.0 * W * H * B
        result["roofline_blend_per_stream_bg"] = {"kernel": "blend", "bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                  "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "avg_ms": round(ms, 4),
                                                  "algorithmic_bytes_per_launch": int(by),
                                                  "note": "bsx_composite_batch with a separate background image per stream: every byte of the 10 B/px is HBM traffic"}
        del d_bg_ps

    main_samples = None
    if rank == 0:
        main_samples = (res["model_path"], res["host"][:4].copy(), res["bg_host"], res["masks_k"], res["out_k"], res["photo"])
    release(res)

    if rank == 0:
        if world > 1:
            result["second_device_check"] = run_second_device_check()
            if c4_samples is not None and not args.no_cpu_baseline and "configs4" in result:
                mp_, fr_, bg_, mk_, out_, photo_ = c4_samples
                result["configs4"]["parity_sample"] = parity_sample(mp_, 1280, 720, fr_, bg_, mk_, out_, need_person=photo_)
        if not args.no_cpu_baseline:                        # kept at N > 1: rank 0, after the other ranks have left (their GPUs idle, the host cores free)
            try:
                result["cpu_baseline"] = cpu_baseline(main_samples[0], W, H, args.cpu_seconds)
                if not args.per_stream_bg:
                    mp_, fr_, bg_, mk_, out_, photo_ = main_samples
                    result["cpu_baseline"]["parity_sample"] = parity_sample(mp_, W, H, fr_, bg_, mk_, out_, need_person=photo_)
            except Exception as e:  # the baseline must never take the GPU number down with it
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        if world == 1 and default_job and not args.no_extra_configs:
            # the other single-GPU BASELINE configurations, same protocol with fewer steps (they are 3-70x longer per step)
            extra_cfgs = [
                ("configs[2]", "batch=256 1280x720 frames, %s" % NAMES["mlkit"], dict(model_key="mlkit", W=1280, H=720, B=256)),
                ("configs[3]", "batch=1024 640x480 frames, %s, animated background: per-step H2D upload of a 480x360 frame from a pinned 36-frame ring + GPU resize "
                               "(grab_background) inside the timed region" % NAMES["deeplab"], dict(model_key="deeplab", W=640, H=480, B=1024, bg_ring=True)),
                ("configs[4] per-GPU slice", "batch=1024 1280x720 frames, %s (8192 streams / 8 GPUs)" % NAMES["full"], dict(model_key="full", W=1280, H=720, B=1024)),
            ]
            result["configs"] = []
            for tag, desc, kw in extra_cfgs:
                try:
                    n_steps = max(3, args.steps // 4)
                    r = measure(steps=n_steps, warmup=2, rank=0, world=1, local_rank=local_rank, profile_iters=2, **kw)
                    frag = summarize(r, load_pmc(kw["B"], kw["W"], kw["H"], r["model_name"]))
                    frag = {"baseline_config": tag, "workload": desc, "steps": n_steps, "warmup": 2, **frag}
                    samples = (r["model_path"], r["host"][:2].copy(), r["bg_host"], r["masks_k"], r["out_k"], r["photo"])
                    release(r)
                    if not args.no_cpu_baseline:
                        mp_, fr_, bg_, mk_, out_, photo_ = samples
                        frag["parity_sample"] = parity_sample(mp_, kw["W"], kw["H"], fr_, bg_, mk_, out_, need_person=photo_)
                    result["configs"].append(frag)
                except Exception as e:
                    result["configs"].append({"baseline_config": tag, "workload": desc, "error": repr(e)})
            # g1: the reduced-precision GEMM modes of the per-launch (DeepLab) path next to the default, each with its own parity sample — the
            # default (split-f16 MFMA, f32-grade) is the one `configs[3]` reports; "fast" (plain f16 operands) is opt-in and IoU-gated
            result["gemm_modes"] = []
            for mode, what in (("fast16", "fast + the depthwise outputs of the fused inverted-residual blocks stored as f16 (16-bit activation storage for the largest tensors that reach HBM)"),
                               ("fast", "plain f16 MFMA operands (1 term), f32 accumulate — what SetAllowFp16PrecisionForFp32 permits (lib/libbackscrub.cc:225)"),
                               ("off", "f32 MFMA (v_mfma_f32_16x16x4_f32), no fused expand+depthwise kernels")):
                try:
                    os.environ["BSX_F16_GEMM"] = mode
                    r = measure(steps=3, warmup=1, rank=0, world=1, local_rank=local_rank, profile_iters=1, model_key="deeplab", W=640, H=480, B=1024, bg_ring=True)
                    frag = {"BSX_F16_GEMM": mode, "what": what, "workload": "configs[3] geometry", "value": round(r["fps"], 1), "unit": "frames/s",
                            "ms_per_step": round(r["ms_per_step"], 4)}
                    samples = (r["model_path"], r["host"][:2].copy(), r["bg_host"], r["masks_k"], r["out_k"], r["photo"])
                    release(r)
                    if not args.no_cpu_baseline:
                        mp_, fr_, bg_, mk_, out_, photo_ = samples
                        frag["parity_sample"] = parity_sample(mp_, 640, 480, fr_, bg_, mk_, out_, need_person=photo_)
                    result["gemm_modes"].append(frag)
                except Exception as e:
                    result["gemm_modes"].append({"BSX_F16_GEMM": mode, "error": repr(e)})
                finally:
                    os.environ.pop("BSX_F16_GEMM", None)
            # g1 for Meet / MLKit: 16-bit activation STORAGE (BSX_ACT16=1: the tensors that cross kernel boundaries or spill out of LDS as halves, f32
            # arithmetic), next to the f32 default that `value` / `configs[2]` report — opt-in, each with its own parity sample
            result["act_modes"] = []
            for tag, kw in (("configs[1]", dict(model_key="lite", W=640, H=480, B=256)), ("configs[2]", dict(model_key="mlkit", W=1280, H=720, B=256))):
                try:
                    os.environ["BSX_ACT16"] = "1"
                    n_steps = max(3, args.steps // 4)
                    r = measure(steps=n_steps, warmup=3, rank=0, world=1, local_rank=local_rank, profile_iters=2, **kw)
                    frag = {"BSX_ACT16": 1, "what": "activation tensors in HBM stored as f16 (A, b0, B, c0, lo2, lo + the middle program's spilled tensors); arithmetic f32",
                            "workload": tag + " geometry", "value": round(r["fps"], 1), "unit": "frames/s", "ms_per_step": round(r["ms_per_step"], 4), "steps": n_steps,
                            "top_launches": [{"name": t["name"], "ms": round(t["avg_ms"], 4)} for t in sorted(r["stats"], key=lambda t: -t["avg_ms"])[:6]]}
                    samples = (r["model_path"], r["host"][:2].copy(), r["bg_host"], r["masks_k"], r["out_k"], r["photo"])
                    release(r)
                    if not args.no_cpu_baseline:
                        mp_, fr_, bg_, mk_, out_, photo_ = samples
                        frag["parity_sample"] = parity_sample(mp_, kw["W"], kw["H"], fr_, bg_, mk_, out_, need_person=photo_)
                    result["act_modes"].append(frag)
                except Exception as e:
                    result["act_modes"].append({"BSX_ACT16": 1, "workload": tag + " geometry", "error": repr(e)})
                finally:
                    os.environ.pop("BSX_ACT16", None)
            try:
                result["single_stream"] = {"what": "bsx_process_host per call (= bs_maskgen_process through the C++ shim): H2D frame, whole mask pipeline, D2H mask, synchronous",
                                           "runs": [single_stream_latency("lite", 640, 480, 200), single_stream_latency("deeplab", 640, 480, 60)],
                                           "published_context_not_measured_here": "reference README.md:177 ~10 FPS DeepLab on two i5 cores; Meet model card ~120 FPS inference on a laptop CPU"}
            except Exception as e:
                result["single_stream"] = {"error": repr(e)}
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
