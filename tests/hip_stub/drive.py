"""Runs UNDER LD_PRELOAD=libhipstub.so (tests/test_device_order.py starts it): drives every GPU-touching entry point of libbsx.so for a context on device 1
while the caller's current device is 0, through the library's real host code.  No torch, no GPU.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from backscrub_amd import api  # noqa: E402  (module import only: api.lib() would pull torch in)


def load():
    L = C.CDLL(api.lib_path())
    for name, res, args in api.SYMBOLS:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    return L


def main():
    model, W, H, n, dev = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    stub = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhipstub.so"))
    L = load()
    calls = []

    def caller_device_after(what):
        calls.append((what, stub.bsx_stub_current_device()))

    msgs = []
    dbg = api.DEBUG_FN(lambda c, m: msgs.append(m.decode(errors="replace")))
    ctx = L.bsx_new(model.encode(), 2, W, H, n, dev, dbg, api.STAGE_FN(), api.STAGE_FN(), api.STAGE_FN(), None)
    caller_device_after("bsx_new")
    if not ctx:
        print(json.dumps({"error": "bsx_new failed: %s" % msgs}))
        return
    info = api._Info()
    L.bsx_get_info(ctx, C.byref(info))
    # "device" buffers are host memory under the stub
    frames = np.zeros((n, H, W, 3), np.uint8)
    bg = np.zeros((H, W, 3), np.uint8)
    out = np.zeros((n, H, W, 3), np.uint8)
    out2 = np.zeros((n, H, W, 2), np.uint8)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    rc = {}
    rc["reset"] = L.bsx_reset(ctx, None); caller_device_after("bsx_reset")
    rc["process_batch"] = L.bsx_process_batch(ctx, p(frames), n, None, None); caller_device_after("bsx_process_batch")
    rc["step"] = L.bsx_step_batch(ctx, p(frames), p(bg), 0, p(out), n, None); caller_device_after("bsx_step_batch")
    rc["step_yuyv"] = L.bsx_step_batch_yuyv(ctx, p(frames), p(bg), 0, p(out2), n, None); caller_device_after("bsx_step_batch_yuyv")
    rc["step_flip"] = L.bsx_step_batch_ex(ctx, p(frames), p(bg), 0, p(out), n, None, 2 | 4); caller_device_after("bsx_step_batch_ex(flip)")
    rc["step_inplace_flip"] = L.bsx_step_batch_ex(ctx, p(frames), p(bg), 0, p(frames), n, None, 2); caller_device_after("bsx_step_batch_ex(in place)")
    rc["step_bgblur"] = L.bsx_step_batch_ex(ctx, p(frames), None, 0, p(out), n, None, 25 << 8); caller_device_after("bsx_step_batch_ex(bgblur)")
    for i in range(3):                       # two-deep pipeline: the first call only enqueues, the next ones fork the composite onto the context's own stream
        rc["step_pipelined_%d" % i] = L.bsx_step_batch_pipelined(ctx, p(frames), p(bg), 0, p(out), n, None, 0); caller_device_after("bsx_step_batch_pipelined")
    rc["step_while_pending_refused"] = 0 if L.bsx_step_batch(ctx, p(frames), p(bg), 0, p(out), n, None) == -1 else -1; caller_device_after("bsx_step_batch(pending)")
    _st = (api.LaunchStat * (info.n_steps + 8))()
    rc["profile_while_pending_refused"] = 0 if L.bsx_profile_batch(ctx, p(frames), p(bg), 0, p(out), n, 1, _st, info.n_steps + 8, None) == -1 else -1; caller_device_after("bsx_profile_batch(pending)")
    rc["stage_while_pending_refused"] = 0 if all(L.bsx_debug_run_stage(ctx, st_, p(frames), n, None) == -1 for st_ in (1, 2, 3)) else -1; caller_device_after("bsx_debug_run_stage(pending)")
    rc["step_pipelined_flush"] = L.bsx_step_batch_pipelined(ctx, None, None, 0, None, 0, None, 0); caller_device_after("bsx_step_batch_pipelined(flush)")
    # argument checks of the pipelined entry point (each must refuse with BSX_EINVAL = -1 and leave nothing pending)
    rc["pipelined_refuses_bgblur"] = 0 if L.bsx_step_batch_pipelined(ctx, p(frames), p(bg), 0, p(out), n, None, 25 << 8) == -1 else -1
    rc["pipelined_refuses_in_place"] = 0 if L.bsx_step_batch_pipelined(ctx, p(frames), p(bg), 0, p(frames), n, None, 0) == -1 else -1
    rc["pipelined_refuses_too_many_streams"] = 0 if L.bsx_step_batch_pipelined(ctx, p(frames), p(bg), 0, p(out), n + 1, None, 0) == -1 else -1
    rc["step_after_refusals"] = L.bsx_step_batch(ctx, p(frames), p(bg), 0, p(out), n, None); caller_device_after("bsx_step_batch(after refusals)")
    rc["pipelined_then_reset_drops_pending"] = L.bsx_step_batch_pipelined(ctx, p(frames), p(bg), 0, p(out), n, None, 0) or L.bsx_reset(ctx, None) or L.bsx_step_batch(ctx, p(frames), p(bg), 0, p(out), n, None)
    caller_device_after("bsx_reset(pending)")
    rc["composite"] = L.bsx_composite_batch(ctx, p(bg), 0, p(frames), None, p(out), n, None); caller_device_after("bsx_composite_batch")
    mask = np.zeros((H, W), np.uint8)
    for i in range(3):                       # first call captures the slot's hipGraph, the next ones replay it
        rc["process_host_%d" % i] = L.bsx_process_host(ctx, 0, p(frames[0]), W * 3, p(mask), W); caller_device_after("bsx_process_host")
    small = np.zeros((1, 360, 480, 3), np.uint8)
    rc["resize"] = L.bsx_resize_bgr(ctx, p(small), 480, 360, p(out), W, H, 1, None); caller_device_after("bsx_resize_bgr")
    rc["yuyv"] = L.bsx_bgr_to_yuyv(ctx, p(out), p(out2), W, H, n, None); caller_device_after("bsx_bgr_to_yuyv")
    rc["yuyv_to_bgr"] = L.bsx_yuyv_to_bgr(ctx, p(out2), p(out), W, H, n, None); caller_device_after("bsx_yuyv_to_bgr")
    rc["flip"] = L.bsx_flip_bgr(ctx, p(frames), p(out), W, H, n, 1, None); caller_device_after("bsx_flip_bgr")
    rc["gauss"] = L.bsx_gaussian_blur_bgr(ctx, p(frames), p(out), W, H, n, 25, None); caller_device_after("bsx_gaussian_blur_bgr")
    stats = (api.LaunchStat * (info.n_steps + 8))()
    rc["profile"] = min(0, L.bsx_profile_batch(ctx, p(frames), p(bg), 0, p(out), n, 1, stats, info.n_steps + 8, None)); caller_device_after("bsx_profile_batch")
    for stage in range(4):
        rc["stage_%d" % stage] = L.bsx_debug_run_stage(ctx, stage, p(frames), n, None); caller_device_after("bsx_debug_run_stage")
    st4 = (C.c_long * 4)()
    rc["tile_stats"] = L.bsx_debug_mask_tile_stats(ctx, n, st4); caller_device_after("bsx_debug_mask_tile_stats")
    ticks = (C.c_ulonglong * 300)()
    rc["timeline"] = min(0, L.bsx_debug_program_timeline(ctx, n, ticks, 300, None)); caller_device_after("bsx_debug_program_timeline")
    # background source + live worker (their own streams / threads)
    ring = np.zeros((3, 36, 48, 3), np.uint8)
    b = L.bsx_background_from_frames(ctx, p(ring), 48, 36, 3, 30.0, 0); caller_device_after("bsx_background_from_frames")
    rc["bg_new"] = 0 if b else -1
    if b:
        rc["bg_grab"] = min(0, L.bsx_background_grab(b, W, H, p(out[0]), None)); caller_device_after("bsx_background_grab")
        L.bsx_background_free(b); caller_device_after("bsx_background_free")
    lv = L.bsx_live_new(ctx); caller_device_after("bsx_live_new")
    rc["live_new"] = 0 if lv else -1
    if lv:
        for _ in range(3):
            rc["live_set"] = L.bsx_live_set_input_frame(lv, p(frames[0]), W * 3); caller_device_after("bsx_live_set_input_frame")
            rc["live_get"] = min(0, L.bsx_live_get_output_mask(lv, p(mask), W)); caller_device_after("bsx_live_get_output_mask")
        L.bsx_live_delete(lv); caller_device_after("bsx_live_delete")
    plan = L.bsx_plan_describe(ctx).decode()
    L.bsx_delete(ctx); caller_device_after("bsx_delete")
    print(json.dumps({"rc": rc, "caller_device": calls, "device_in_info": info.device, "n_steps": info.n_steps,
                      "specialised": "specialised kernel" in plan, "messages": msgs}))


if __name__ == "__main__":
    main()
