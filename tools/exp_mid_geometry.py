"""Round 6 experiment: the middle kernel's workgroup geometry — BSX_MID_LANES x BSX_MID_LDS_KB, set by the CALLER's environment — on one bench
configuration: the synchronous step, the two-deep pipelined step, and the middle kernel's own duration.  One line per run; tools/calls/r06_call2.sh alternates the
variants on one box.

    BSX_MID_LANES=512 BSX_MID_LDS_KB=80 python tools/exp_mid_geometry.py --model lite --batch 256 [--width 640 --height 480] [--steps 100]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="lite")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import torch

    import bench
    res = bench.measure(a.model, a.width, a.height, a.batch, a.steps, 10, 0, 1, 0, profile_iters=4, ramp_s=1.0, static_leg=False, parity_streams=2, parity_steps=4)
    mg, ring, d_bg = res["mg"], res["frames_ring"], res["d_bg"]
    T = len(ring)
    outs = [torch.empty_like(res["d_out"]) for _ in range(2)]
    mg.reset()
    for t in range(10):
        mg.step_pipelined(ring[t % T], d_bg, outs[t & 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(a.steps):
        mg.step_pipelined(ring[t % T], d_bg, outs[t & 1])
    torch.cuda.synchronize()
    ms_pipe = 1e3 * (time.perf_counter() - t0) / a.steps
    mg.flush_pipelined()
    torch.cuda.synchronize()
    par = bench.parity_sequence(res["model_path"], a.width, a.height, res["parity_in"])
    st = {s["name"]: round(s["avg_ms"] * 1e3, 1) for s in res["stats"]}
    print(json.dumps({"tag": a.tag, "lanes": os.environ.get("BSX_MID_LANES", "1024"), "lds_kb": os.environ.get("BSX_MID_LDS_KB", "160"), "model": a.model, "batch": a.batch,
                      "frame": "%dx%d" % (a.width, a.height), "step_ms": round(res["ms_per_step"], 4), "fps": round(res["fps"]), "pipelined_ms": round(ms_pipe, 4),
                      "pipelined_fps": round(a.batch / ms_pipe * 1e3), "mid_us": st.get("frame_program"), "launch_us": st,
                      "iou_min": par.get("mask_iou_min"), "max_abs": par.get("composite_max_abs_diff")}), flush=True)
    bench.release(res)


if __name__ == "__main__":
    main()
