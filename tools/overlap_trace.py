"""Do the two halves of bsx_step_batch_pipelined really overlap, and what does the overlap cost each kernel?  (run ON the GPU box)

    rocprofv3 --kernel-trace -d gpurun_out/ovl -o t -- python tools/overlap_trace.py --run lite 640 480 256
    python tools/overlap_trace.py --summary gpurun_out/ovl/*/t_results.db   (or wherever rocprofv3 put the rocpd database)

--run: 30 synchronous steps, one flip_bgr_k launch as a separator, 30 pipelined steps.  --summary: per kernel the average duration in both phases, and for the
pipelined phase how much of the mask-tile launch ran while a kernel of the other stream was running (from the start / end stamps of the dispatches).
"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(key, W, H, B):
    import torch

    import backscrub_amd
    from backscrub_amd import synth
    from bench import resolve_model
    path = resolve_model(key)[0]
    mg = backscrub_amd.MaskGen(path, W, H, n_streams=B)
    host = synth.frames(16, W, H)
    d_frames = torch.from_numpy(host).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
    d_bg = torch.from_numpy(synth.background(W, H)).cuda()
    out = torch.empty_like(d_frames)
    for _ in range(30):                      # warm: clocks, temporal state
        mg.step(d_frames, d_bg, out)
    torch.cuda.synchronize()
    for _ in range(30):
        mg.step(d_frames, d_bg, out)
    torch.cuda.synchronize()
    mg.flip_bgr(d_frames[:1], 1)             # separator launch (flip_bgr_k)
    torch.cuda.synchronize()
    for _ in range(31):
        mg.step_pipelined(d_frames, d_bg, out)
    mg.flush_pipelined()
    torch.cuda.synchronize()
    mg.close()


def summary(paths):
    from tools.rocpd_summary import short
    for path in paths:
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
        qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
        if "start" not in cols or "end" not in cols:
            print("kernels view has no start / end columns:", cols)
            continue
        rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
        rows = [(short(n), s, e, q) for n, s, e, q in rows]
        sep = [i for i, r in enumerate(rows) if r[0].startswith("flip_bgr_k")]
        if not sep:
            print("no separator launch found")
            continue
        # the last 30 synchronous steps end at the separator; the pipelined steps follow it
        ours = lambda r: r[0].startswith(("prep_fused_k", "seg_", "bsx_mid", "mask_tile_k", "tile_class_k", "outside_roi", "bsx::tile_class_k", "ir_expand", "pw_gemm", "dl_head0", "resize_argmax"))  # noqa: E731
        a = [r for r in rows[:sep[-1]] if ours(r)]
        b = [r for r in rows[sep[-1] + 1:] if ours(r)]
        n_step = len([r for r in b if r[0].startswith("prep_fused_k")])
        a = a[-(len(b)):] if len(a) > len(b) else a
        print("%s: %d dispatches synchronous (tail of the phase), %d pipelined (%d steps); queue column: %s" % (path, len(a), len(b), n_step, qcol))
        names = []
        for r in a + b:
            if r[0] not in names:
                names.append(r[0])
        print("| kernel | sync avg us | pipelined avg us | ratio | queues (pipelined) |")
        print("|---|---:|---:|---:|---|")
        for n in names:
            da = [(e - s) / 1e3 for k, s, e, _ in a if k == n]
            dbb = [(e - s) / 1e3 for k, s, e, _ in b if k == n]
            qs = sorted({str(q) for k, _, _, q in b if k == n})
            if da and dbb:
                print("| %s | %.2f | %.2f | %.2f | %s |" % (n, sum(da) / len(da), sum(dbb) / len(dbb), (sum(dbb) / len(dbb)) / (sum(da) / len(da)), ",".join(qs)))
        span = lambda rs: (max(r[2] for r in rs) - min(r[1] for r in rs)) / 1e3 if rs else 0.0  # noqa: E731
        steps_a = len([r for r in a if r[0].startswith("prep_fused_k")])
        print("wall per step: synchronous %.2f us (%d steps), pipelined %.2f us (%d steps)" % (span(a) / max(steps_a, 1), steps_a, span(b) / max(n_step, 1), n_step))
        print("sum of kernel durations per step: synchronous %.2f us, pipelined %.2f us" % (sum(r[2] - r[1] for r in a) / 1e3 / max(steps_a, 1), sum(r[2] - r[1] for r in b) / 1e3 / max(n_step, 1)))
        # overlap: time of each mask_tile_k dispatch of the pipelined phase during which some other-kernel dispatch was running
        others = sorted((s, e) for k, s, e, _ in b if not k.startswith(("mask_tile_k", "tile_class_k", "bsx::tile_class_k", "outside_roi")))
        tot = ov = 0.0
        for k, s, e, _ in b:
            if not k.startswith("mask_tile_k"):
                continue
            tot += e - s
            for s2, e2 in others:
                if e2 <= s:
                    continue
                if s2 >= e:
                    break
                ov += min(e, e2) - max(s, s2)
        print("mask_tile_k (pipelined phase): %.1f %% of its run time overlapped a kernel of the other stream" % (100.0 * ov / max(tot, 1.0)))


if __name__ == "__main__":
    if len(sys.argv) >= 6 and sys.argv[1] == "--run":
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    elif len(sys.argv) >= 3 and sys.argv[1] == "--summary":
        summary([p for a in sys.argv[2:] for p in (glob.glob(a) or [a])])
    else:
        print(__doc__)
