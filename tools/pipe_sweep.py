"""bsx_step_batch_pipelined: occupancy cap of the composite (BSX_PIPE_WGS) x stream priority (BSX_PIPE_PRIO) against the synchronous step — one box, one process.
    python tools/pipe_sweep.py [lite|mlkit|full ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"lite": (640, 480, 256, 200), "mlkit": (1280, 720, 256, 60), "full": (1280, 720, 1024, 30), "deeplab": (640, 480, 1024, 8)}


def main(keys):
    import torch

    import backscrub_amd
    from backscrub_amd import synth
    from bench import resolve_model
    for key in keys:
        W, H, B, steps = CFG[key]
        path = resolve_model(key)[0]
        host = synth.frames(16, W, H)
        d_frames = torch.from_numpy(host).cuda().repeat((B + 15) // 16, 1, 1, 1)[:B].contiguous()
        d_bg = torch.from_numpy(synth.background(W, H)).cuda()
        out = torch.empty_like(d_frames)

        def timed(mg, fn, n):
            for _ in range(max(5, n // 4)):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
            return best * 1e3

        mg = backscrub_amd.MaskGen(path, W, H, n_streams=B)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:          # clock ramp
            mg.step(d_frames, d_bg, out)
        torch.cuda.synchronize()
        sync_ms = timed(mg, lambda: mg.step(d_frames, d_bg, out), steps)
        ref = out.clone()
        mg.close()
        print("%s %dx%d x%d: synchronous %.4f ms/step (%.0f fps)" % (key, W, H, B, sync_ms, B / sync_ms * 1e3), flush=True)
        for prio in ("1", "0"):
            for wgs in ("0", "2", "3", "4", "5"):
                os.environ["BSX_PIPE_WGS"] = wgs
                os.environ["BSX_PIPE_PRIO"] = prio
                mg = backscrub_amd.MaskGen(path, W, H, n_streams=B)
                for _ in range(30):
                    mg.step(d_frames, d_bg, out)
                ms = timed(mg, lambda: mg.step_pipelined(d_frames, d_bg, out), steps)
                mg.flush_pipelined()
                torch.cuda.synchronize()
                same = bool(torch.equal(out, ref))
                mg.close()
                print("  low_prio=%s wgs_per_cu=%s: %.4f ms/step  x%.3f  identical=%s" % (prio, wgs if wgs != "0" else "uncapped", ms, sync_ms / ms, same), flush=True)
        del d_frames, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(sys.argv[1:] or ["lite", "mlkit", "full"])
