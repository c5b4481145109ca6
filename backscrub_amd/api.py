"""ctypes binding of libbsx.so and the Python mirror of the reference interface.

Reference interface mirrored here (same names / argument meaning / error behaviour):
  lib/libbackscrub.h:13   bs_tensorflow_version()
  lib/libbackscrub.h:16   bs_maskgen_new(modelname, threads, width, height, ondebug, onprep, oninfer, onmask, caller_ctx)
  lib/libbackscrub.h:36   bs_maskgen_delete(context)
  lib/libbackscrub.h:39   bs_maskgen_process(context, frame, mask) -> bool
  app/deepseg.cc:108      alpha_blend(srca, srcb, mask)

PyTorch is used only as the owner of device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u8p = C.POINTER(C.c_uint8)
DEBUG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
STAGE_FN = C.CFUNCTYPE(None, C.c_void_p)


class BsxError(RuntimeError):
    pass


class _Info(C.Structure):
    _fields_ = [("model_type", C.c_int), ("width", C.c_int), ("height", C.c_int), ("n_streams", C.c_int),
                ("in_w", C.c_int), ("in_h", C.c_int), ("in_c", C.c_int), ("out_w", C.c_int), ("out_h", C.c_int),
                ("out_c", C.c_int), ("roi", C.c_int * 4), ("in_roi", C.c_int * 4), ("n_ops", C.c_int), ("n_steps", C.c_int),
                ("device", C.c_int), ("norm_scale", C.c_float), ("norm_offset", C.c_float), ("nn_flops_per_frame", C.c_double),
                ("act_bytes_per_stream", C.c_size_t)]


class LaunchStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("avg_ms", C.c_double), ("bytes", C.c_double), ("flops", C.c_double)]


# every symbol include/bsx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bsx_version", C.c_char_p, []),
    ("bsx_device_count", C.c_int, []),
    ("bsx_new", C.c_void_p, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, DEBUG_FN, STAGE_FN, STAGE_FN, STAGE_FN, C.c_void_p]),
    ("bsx_delete", None, [C.c_void_p]),
    ("bsx_get_info", C.c_int, [C.c_void_p, C.POINTER(_Info)]),
    ("bsx_last_error", C.c_char_p, [C.c_void_p]),
    ("bsx_reset", C.c_int, [C.c_void_p, C.c_void_p]),
    ("bsx_process_host", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    ("bsx_process_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("bsx_masks_device", C.c_void_p, [C.c_void_p]),
    ("bsx_composite_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("bsx_step_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    ("bsx_step_batch_yuyv", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    ("bsx_step_batch_ex", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_uint]),
    ("bsx_step_batch_pipelined", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_uint]),
    ("bsx_resize_bgr", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("bsx_bgr_to_yuyv", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("bsx_yuyv_to_bgr", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("bsx_background_load", C.c_void_p, [C.c_void_p, C.c_char_p, C.c_int]),
    ("bsx_background_from_frames", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    ("bsx_background_free", None, [C.c_void_p]),
    ("bsx_background_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("bsx_background_grab", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("bsx_media_decode", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p, C.c_size_t]),
    ("bsx_media_free", None, [C.POINTER(C.c_uint8)]),
    ("bsx_live_new", C.c_void_p, [C.c_void_p]),
    ("bsx_live_delete", None, [C.c_void_p]),
    ("bsx_live_set_input_frame", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("bsx_live_get_output_mask", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("bsx_live_timings", C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    ("bsx_gaussian_blur_bgr", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("bsx_debug_gauss_coeffs", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("bsx_flip_bgr", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("bsx_debug_buffer", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("bsx_debug_run_stage", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    ("bsx_plan_describe", C.c_char_p, [C.c_void_p]),
    ("bsx_debug_tensor", C.c_long, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_long]),
    ("bsx_model_precompile", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("bsx_model_kernel_source", C.c_long, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("bsx_model_seg_source", C.c_long, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("bsx_debug_tensor_of", C.c_long, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_long]),
    ("bsx_debug_mask_tile_stats", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_long)]),
    ("bsx_debug_program_timeline", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int, C.c_void_p]),
    ("bsx_model_describe", C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("bsx_profile_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.POINTER(LaunchStat), C.c_int, C.c_void_p]),
]


def lib_path() -> str:
    return os.environ.get("BSX_LIBRARY") or os.path.join(_HERE, "libbsx.so")   # BSX_LIBRARY: A/B another build of the same ABI


def lib():
    """Load libbsx.so.  Fails loudly (no fallback) when it has not been built."""
    global _LIB
    if _LIB is None:
        # torch bundles its own HIP runtime under the same SONAME as /opt/rocm's; importing it first makes the
        # whole process (torch allocations + our launches) use ONE runtime instance.
        import torch  # noqa: F401
        p = lib_path()
        if not os.path.exists(p):
            raise BsxError("libbsx.so is missing (%s): run `python -m backscrub_amd.build` — there is no CPU fallback" % p)
        L = C.CDLL(p)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)  # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def _check(rc, ctx=None, what=""):
    if rc != 0:
        msg = lib().bsx_last_error(ctx) or b""
        raise BsxError("%s failed (%d): %s" % (what, rc, msg.decode(errors="replace").strip()))


def _torch():
    import torch
    return torch


def _stream_ptr():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MaskGen:
    """Batched mask generator: n_streams independent camera streams on one GPU."""

    def __init__(self, model_path, width, height, n_streams=1, device=0, threads=2, ondebug=None, onprep=None, oninfer=None,
                 onmask=None, caller_ctx=None):
        L = lib()
        self._cbs = (DEBUG_FN(ondebug) if ondebug else DEBUG_FN(), STAGE_FN(onprep) if onprep else STAGE_FN(),
                     STAGE_FN(oninfer) if oninfer else STAGE_FN(), STAGE_FN(onmask) if onmask else STAGE_FN())
        self.h = L.bsx_new(os.fsencode(model_path), threads, width, height, n_streams, device, *self._cbs, caller_ctx)
        if not self.h:
            raise BsxError("bsx_new failed: %s" % (L.bsx_last_error(None) or b"").decode(errors="replace").strip())
        info = _Info()
        _check(L.bsx_get_info(self.h, C.byref(info)), self.h, "bsx_get_info")
        self.info = {k: (list(getattr(info, k)) if k in ("roi", "in_roi") else getattr(info, k)) for k, _ in _Info._fields_}
        self.width, self.height, self.n_streams, self.device = width, height, n_streams, device

    # ---- batched device path -----------------------------------------------------------------
    def process_batch(self, frames, masks_out=None):
        """frames: torch u8 cuda [n,H,W,3] contiguous.  Returns a torch view [n,H,W] of the persistent masks."""
        n = self._n(frames)
        mo = C.c_void_p(masks_out.data_ptr()) if masks_out is not None else None
        _check(lib().bsx_process_batch(self.h, C.c_void_p(frames.data_ptr()), n, mo, _stream_ptr()), self.h, "bsx_process_batch")
        return self.masks()[:n]

    def composite(self, bg, frames, masks=None, out=None):
        torch = _torch()
        n = self._n(frames)
        if out is None:
            out = torch.empty_like(frames)
        stride = self._bg(bg, n)
        mp = C.c_void_p(masks.data_ptr()) if masks is not None else None
        _check(lib().bsx_composite_batch(self.h, C.c_void_p(bg.data_ptr()), stride, C.c_void_p(frames.data_ptr()), mp,
                                         C.c_void_p(out.data_ptr()), n, _stream_ptr()), self.h, "bsx_composite_batch")
        return out

    def _bg(self, bg, n):
        """background operand of a step → bg_frame_stride: ONE image [H,W,3] shared by all streams, or one per stream [>=n,H,W,3] — cuda uint8 on the context's
        device with a contiguous [H,W,3] inside; the stream dimension may be strided or expanded (stride 0 = one image for all), which is what the C ABI's
        bg_frame_stride serves.  (every entry point validates, not only step_pipelined.)"""
        ok = (bg.dtype == _torch().uint8 and bg.is_cuda and bg.dim() in (3, 4) and tuple(bg.shape[-3:]) == (self.height, self.width, 3)
              and tuple(bg.stride()[-3:]) == (self.width * 3, 3, 1) and (bg.dim() == 3 or bg.shape[0] >= n) and bg.device.index == self.device
              and (bg.dim() == 3 or bg.stride(0) >= 0))
        if not ok:
            raise BsxError("bg must be a cuda:%d uint8 tensor [%d,%d,3] or [>=%d,%d,%d,3] with contiguous images" % (self.device, self.height, self.width, n, self.height, self.width))
        return 0 if bg.dim() == 3 else int(bg.stride(0))

    def step(self, frames, bg, out):
        n = self._n(frames)
        stride = self._bg(bg, n)
        _check(lib().bsx_step_batch(self.h, C.c_void_p(frames.data_ptr()), C.c_void_p(bg.data_ptr()), stride,
                                    C.c_void_p(out.data_ptr()), n, _stream_ptr()), self.h, "bsx_step_batch")
        return out

    def step_yuyv(self, frames, bg, out_yuyv):
        """one main-loop iteration with the composite written as YUYV 4:2:2 [n,H,W,2] (convert_rgb_to_yuyv fused into the blend)"""
        n = self._n(frames)
        if (out_yuyv.dim() != 4 or tuple(out_yuyv.shape[1:]) != (self.height, self.width, 2) or out_yuyv.shape[0] < n or not out_yuyv.is_contiguous()
                or not out_yuyv.is_cuda or out_yuyv.dtype != _torch().uint8):
            raise BsxError("out_yuyv must be a contiguous cuda uint8 tensor [>=%d,%d,%d,2]" % (n, self.height, self.width))
        stride = self._bg(bg, n)
        _check(lib().bsx_step_batch_yuyv(self.h, C.c_void_p(frames.data_ptr()), C.c_void_p(bg.data_ptr()), stride,
                                         C.c_void_p(out_yuyv.data_ptr()), n, _stream_ptr()), self.h, "bsx_step_batch_yuyv")
        return out_yuyv

    def step_ex(self, frames, bg, out, flip_h=False, flip_v=False, yuyv=False, no_mask=False, bgblur=0, yuyv_in=False):
        """one main-loop iteration with cv::flip of the composite (app/deepseg.cc:667-673) and / or the YUYV pack folded into the blend's store;
        bgblur=<odd ksize>: the background is GaussianBlur(the stream's own frame) (-p bgblur:<n> without -b, deepseg.cc:652-661), `bg` may be None;
        yuyv_in: `frames` is the camera's raw YUYV 4:2:2 [n,H,W,2] (cv::COLOR_YUV2BGR_YUYV folded into the kernels that read it: BSX_STEP_YUYV_IN)"""
        n = self._n(frames, yuyv_in)
        want = (self.height, self.width, 2 if yuyv else 3)
        if out.dim() != 4 or tuple(out.shape[1:]) != want or out.shape[0] < n or not out.is_contiguous() or not out.is_cuda or out.dtype != _torch().uint8:
            raise BsxError("out must be a contiguous cuda uint8 tensor [>=%d,%d,%d,%d]" % ((n,) + want))
        if bg is None and not bgblur:
            raise BsxError("bg is required unless bgblur is set")
        stride = 0 if bg is None else self._bg(bg, n)
        flags = (1 if yuyv else 0) | (2 if flip_h else 0) | (4 if flip_v else 0) | (8 if no_mask else 0) | (16 if yuyv_in else 0) | ((int(bgblur) & 255) << 8)
        _check(lib().bsx_step_batch_ex(self.h, C.c_void_p(frames.data_ptr()), C.c_void_p(bg.data_ptr() if bg is not None else None), stride,
                                       C.c_void_p(out.data_ptr()), n, _stream_ptr(), flags), self.h, "bsx_step_batch_ex")
        return out

    def step_pipelined(self, frames, bg, out, flip_h=False, flip_v=False, yuyv=False, no_mask=False, yuyv_in=False):
        """throughput mode (bsx_step_batch_pipelined): enqueue the mask pipeline of THIS batch and, concurrently, the composite of the batch handed over by
        the previous call (the reference's CalcMask worker next to its blend loop, app/deepseg.cc:159-285).  `out` — and masks() — hold THIS batch's results
        once the NEXT call (or flush_pipelined()) has completed; frames / bg / out must stay untouched until then.  Bit-identical to step_ex per batch."""
        n = self._n(frames, yuyv_in)
        want = (self.height, self.width, 2 if yuyv else 3)
        if out.dim() != 4 or tuple(out.shape[1:]) != want or out.shape[0] < n or not out.is_contiguous() or not out.is_cuda or out.dtype != _torch().uint8:
            raise BsxError("out must be a contiguous cuda uint8 tensor [>=%d,%d,%d,%d]" % ((n,) + want))
        stride = self._bg(bg, n)
        flags = (1 if yuyv else 0) | (2 if flip_h else 0) | (4 if flip_v else 0) | (8 if no_mask else 0) | (16 if yuyv_in else 0)
        _check(lib().bsx_step_batch_pipelined(self.h, C.c_void_p(frames.data_ptr()), C.c_void_p(bg.data_ptr()), stride, C.c_void_p(out.data_ptr()), n, _stream_ptr(), flags),
               self.h, "bsx_step_batch_pipelined")
        # the composite of THIS batch is enqueued by the NEXT call, on a stream torch's caching allocator knows nothing about: the three tensors stay referenced
        # here until then (a caller that drops them would otherwise have their memory handed out again under the pending kernel) — and for ONE MORE call: a caller
        # that alternates streams gets composite k joined into the stream of call k + 1, so a block of call k's stream freed right after call k + 1 could be handed
        # out again on that stream while the composite still runs.  Released after call k + 2, a flush or a reset.
        held = getattr(self, "_pending", None)
        self._pending_prev = held
        self._pending = (frames, bg, out)
        return out

    def flush_pipelined(self):
        """composite the batch still pending in the two-deep pipeline (on the current stream)"""
        _check(lib().bsx_step_batch_pipelined(self.h, None, None, 0, None, 0, _stream_ptr(), 0), self.h, "bsx_step_batch_pipelined(flush)")
        self._release_pending()

    def _release_pending(self):
        """drop the references step_pipelined holds: after a flush the composite that used them is ordered on the caller's own stream, after a reset it was dropped
        with the state.  (Between calls k and k + 1 the references move on by themselves: call k + 1 orders the caller's stream behind composite k — the launch that
        advances the temporal state waits for it — so memory freed after that call cannot be reused ahead of the composite.)"""
        self._pending = None
        self._pending_prev = None

    def profile(self, frames, bg, out, iters=5):
        """per-launch hipEvent timings of the whole per-batch sequence → list of dicts"""
        n = self._n(frames)
        cap = self.info["n_steps"] + 8
        arr = (LaunchStat * cap)()
        stride = self._bg(bg, n)
        k = lib().bsx_profile_batch(self.h, C.c_void_p(frames.data_ptr()), C.c_void_p(bg.data_ptr()), stride, C.c_void_p(out.data_ptr()),
                                    n, iters, arr, cap, _stream_ptr())
        if k < 0:
            _check(k, self.h, "bsx_profile_batch")
        return [dict(name=arr[i].name.decode(), avg_ms=arr[i].avg_ms, bytes=arr[i].bytes, flops=arr[i].flops) for i in range(k)]

    def program_timeline(self, n=None):
        """per-micro-op durations (microseconds, workgroup 0) of the per-frame network program"""
        n = n or self.n_streams
        cap = 1024 + 64 * 16 * 4
        arr = (C.c_ulonglong * cap)()
        k = lib().bsx_debug_program_timeline(self.h, n, arr, cap, _stream_ptr())
        if k < 0:
            _check(k, self.h, "bsx_debug_program_timeline")
        self.last_subphase_us = [arr[256 + i] / 100.0 for i in range(12)]   # debug accumulators of instrumented micro-ops
        # fine[op][wave] = (wait+barrier, weight-DMA issue, body) in shader cycles, lane 0 of each wave of workgroup 0
        self.last_fine = [[tuple(arr[1024 + (i * 16 + w) * 4 + j] for j in range(4)) for w in range(16)] for i in range(min(k, 64))]
        return [(arr[i + 1] - arr[i]) / 100.0 for i in range(k)]

    def masks(self):
        """torch u8 view [n_streams,H,W] of the lib-owned persistent masks (cf. `mask = ctx.mask`, libbackscrub.cc:374)."""
        return self._view(3, "uint8", (self.n_streams, self.height, self.width))

    def reset(self):
        _check(lib().bsx_reset(self.h, _stream_ptr()), self.h, "bsx_reset")
        self._release_pending()

    def resize_bgr(self, src, dw, dh):
        torch = _torch()
        n, sh, sw, _ = src.shape
        dst = torch.empty((n, dh, dw, 3), dtype=torch.uint8, device=src.device)
        _check(lib().bsx_resize_bgr(self.h, C.c_void_p(src.data_ptr()), sw, sh, C.c_void_p(dst.data_ptr()), dw, dh, n, _stream_ptr()),
               self.h, "bsx_resize_bgr")
        return dst

    def bgr_to_yuyv(self, bgr):
        torch = _torch()
        n, h, w, _ = bgr.shape
        out = torch.empty((n, h, w, 2), dtype=torch.uint8, device=bgr.device)
        _check(lib().bsx_bgr_to_yuyv(self.h, C.c_void_p(bgr.data_ptr()), C.c_void_p(out.data_ptr()), w, h, n, _stream_ptr()),
               self.h, "bsx_bgr_to_yuyv")
        return out

    def flip_bgr(self, bgr, code):
        """cv::flip(bgr, out, code) for [n,h,w,3] u8 device frames (deepseg.cc:667-673): 0 vertical, >0 horizontal, <0 both."""
        torch = _torch()
        n, h, w, _ = bgr.shape
        out = torch.empty_like(bgr)
        _check(lib().bsx_flip_bgr(self.h, C.c_void_p(bgr.data_ptr()), C.c_void_p(out.data_ptr()), w, h, n, int(code), _stream_ptr()), self.h, "bsx_flip_bgr")
        return out

    def gaussian_blur(self, bgr, ksize=25, out=None):
        """cv::GaussianBlur(bgr, out, Size(ksize, ksize), 0) for [n,h,w,3] u8 device frames (deepseg.cc:657-658, -p bgblur:<ksize>)."""
        torch = _torch()
        n, h, w, _ = bgr.shape
        if out is None:
            out = torch.empty_like(bgr)
        elif out.shape != bgr.shape or out.dtype != bgr.dtype or not out.is_contiguous() or out.device != bgr.device:
            raise BsxError("out must be a contiguous tensor shaped like the input")
        _check(lib().bsx_gaussian_blur_bgr(self.h, C.c_void_p(bgr.data_ptr()), C.c_void_p(out.data_ptr()), w, h, n, int(ksize), _stream_ptr()),
               self.h, "bsx_gaussian_blur_bgr")
        return out

    def yuyv_to_bgr(self, yuyv):
        torch = _torch()
        n, h, w, _ = yuyv.shape
        out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=yuyv.device)
        _check(lib().bsx_yuyv_to_bgr(self.h, C.c_void_p(yuyv.data_ptr()), C.c_void_p(out.data_ptr()), w, h, n, _stream_ptr()),
               self.h, "bsx_yuyv_to_bgr")
        return out

    # ---- drop-in single frame path (host buffers) -----------------------------------------------
    def process_host(self, frame: np.ndarray, stream_idx=0, mask_out: np.ndarray | None = None) -> np.ndarray:
        if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3 or frame.shape[0] != self.height or frame.shape[1] != self.width:
            raise BsxError("frame must be uint8 [%d,%d,3]" % (self.height, self.width))
        if frame.strides[2] != 1 or frame.strides[1] != 3:
            frame = np.ascontiguousarray(frame)
        if mask_out is None:
            mask_out = np.empty((self.height, self.width), np.uint8)
        elif (not isinstance(mask_out, np.ndarray) or mask_out.dtype != np.uint8 or mask_out.shape != (self.height, self.width)
              or mask_out.strides[1] != 1 or mask_out.strides[0] < self.width or not mask_out.flags.writeable):
            # the C side writes height rows of width bytes: anything else would scribble over the host heap
            raise BsxError("mask_out must be a writable uint8 [%d,%d] array with unit inner stride" % (self.height, self.width))
        _check(lib().bsx_process_host(self.h, stream_idx, frame.ctypes.data, frame.strides[0], mask_out.ctypes.data, mask_out.strides[0]),
               self.h, "bsx_process_host")
        return mask_out

    # ---- introspection (tests) --------------------------------------------------------------------
    def run_stage(self, stage, frames=None, n=None):
        n = n if n is not None else (self._n(frames, yuyv_in=(stage == 4)) if frames is not None else self.n_streams)
        fp = C.c_void_p(frames.data_ptr()) if frames is not None else None
        _check(lib().bsx_debug_run_stage(self.h, stage, fp, n, _stream_ptr()), self.h, "bsx_debug_run_stage")

    def input_tensor(self):
        i = self.info
        return self._view(0, "float32", (self.n_streams, i["in_h"], i["in_w"], i["in_c"]))

    def output_tensor(self):
        i = self.info
        return self._view(1, "float32", (self.n_streams, i["out_h"], i["out_w"], i["out_c"]))

    def ofinal(self):
        i = self.info
        return self._view(2, "uint8", (self.n_streams, i["out_h"], i["out_w"]))

    def plan(self) -> str:
        return lib().bsx_plan_describe(self.h).decode()

    def mask_tile_stats(self, n=None) -> dict:
        """how the fused mask + blend launch classifies its tiles for the current temporal state: a uniform tile (whole source block 0xFF / 0x00) skips the
        mask phases and reads only the operand its composite copies"""
        a = (C.c_long * 4)()
        _check(lib().bsx_debug_mask_tile_stats(self.h, n or self.n_streams, a), self.h, "bsx_debug_mask_tile_stats")
        return {"tiles": a[0], "uniform_255": a[1], "uniform_0": a[2], "general": a[3], "tile": "128x32"}

    def graph_tensor(self, idx, stream=None) -> np.ndarray:
        if stream is not None:
            n = lib().bsx_debug_tensor_of(self.h, idx, stream, None, 0)
            if n < 0:
                raise BsxError("tensor %d not materialised" % idx)
            a = np.empty(n, np.float32)
            lib().bsx_debug_tensor_of(self.h, idx, stream, a.ctypes.data_as(C.POINTER(C.c_float)), n)
            return a
        n = lib().bsx_debug_tensor(self.h, idx, None, 0)
        if n < 0:
            raise BsxError("tensor %d not materialised" % idx)
        _torch().cuda.synchronize()
        a = np.empty(n, np.float32)
        lib().bsx_debug_tensor(self.h, idx, a.ctypes.data_as(C.POINTER(C.c_float)), n)
        return a

    def _view(self, which, dtype, shape):
        torch = _torch()
        p, b = C.c_void_p(), C.c_size_t()
        _check(lib().bsx_debug_buffer(self.h, which, C.byref(p), C.byref(b)), self.h, "bsx_debug_buffer")
        return _as_torch(p.value, b.value, dtype, shape, self.device)

    def _n(self, frames, yuyv_in=False):
        ch = 2 if yuyv_in else 3                 # BSX_STEP_YUYV_IN: the camera's raw 4:2:2 frames, 2 bytes per pixel
        if frames.dim() != 4 or tuple(frames.shape[1:]) != (self.height, self.width, ch) or not frames.is_contiguous() or not frames.is_cuda or frames.dtype != _torch().uint8:
            raise BsxError("frames must be a contiguous cuda uint8 tensor [n,%d,%d,%d]" % (self.height, self.width, ch))
        if frames.shape[0] > self.n_streams:
            raise BsxError("batch larger than n_streams")
        return int(frames.shape[0])

    def close(self):
        if getattr(self, "h", None):
            lib().bsx_delete(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CudaArray:
    """__cuda_array_interface__ holder so torch can alias lib-owned device memory without copying."""

    def __init__(self, ptr, nbytes, typestr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}
        self._nbytes = nbytes


def _as_torch(ptr, nbytes, dtype, shape, device):
    torch = _torch()
    typestr = {"uint8": "|u1", "float32": "<f4"}[dtype]
    return torch.as_tensor(_CudaArray(ptr, nbytes, typestr, shape), device="cuda:%d" % device)


def media_decode(path: str):
    """Host-only decode of a background file (GIF / PNG / PPM) → (frames [n,h,w,3] u8 BGR, fps).  BsxError with the decoder's reason."""
    w, h, fps, ptr = C.c_int(), C.c_int(), C.c_double(), C.POINTER(C.c_uint8)()
    err = C.create_string_buffer(512)
    n = lib().bsx_media_decode(os.fsencode(path), C.byref(w), C.byref(h), C.byref(fps), C.byref(ptr), err, len(err))
    if n <= 0:
        raise BsxError(err.value.decode(errors="replace") or "media decode failed (%d)" % n)
    try:
        frames = np.ctypeslib.as_array(ptr, (n, h.value, w.value, 3)).copy()
    finally:
        lib().bsx_media_free(ptr)
    return frames, fps.value


class Background:
    """load_background / grab_background of app/background.cc on top of a MaskGen's GPU."""

    def __init__(self, mg: MaskGen, path=None, frames=None, fps=0.0, debug=0):
        self.mg = mg
        if path is not None:
            self.h = lib().bsx_background_load(mg.h, os.fsencode(path), debug)
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            n, hh, ww, _ = frames.shape
            self.h = lib().bsx_background_from_frames(mg.h, frames.ctypes.data, ww, hh, n, float(fps), debug)
        if not self.h:
            raise BsxError("cannot load background")
        w, h, n, f, v = C.c_int(), C.c_int(), C.c_int(), C.c_double(), C.c_int()
        lib().bsx_background_info(self.h, C.byref(w), C.byref(h), C.byref(n), C.byref(f), C.byref(v))
        self.width, self.height, self.n_frames, self.fps, self.video = w.value, h.value, n.value, f.value, bool(v.value)

    def grab(self, width, height, out=None):
        """→ (frame number, torch u8 cuda [height,width,3])"""
        torch = _torch()
        if out is None:
            out = torch.empty((height, width, 3), dtype=torch.uint8, device="cuda:%d" % self.mg.device)
        frm = lib().bsx_background_grab(self.h, width, height, C.c_void_p(out.data_ptr()), _stream_ptr())
        if frm < 0:
            raise BsxError("bsx_background_grab failed")
        return frm, out

    def close(self):
        if getattr(self, "h", None):
            lib().bsx_background_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Live:
    """CalcMask (app/deepseg.cc:159-286): set_input_frame / get_output_mask around a worker thread."""

    def __init__(self, mg: MaskGen):
        self.mg = mg
        self.h = lib().bsx_live_new(mg.h)
        if not self.h:
            raise BsxError("bsx_live_new failed")

    def set_input_frame(self, frame: np.ndarray):
        if not isinstance(frame, np.ndarray) or frame.dtype != np.uint8 or frame.shape != (self.mg.height, self.mg.width, 3):
            raise BsxError("frame must be uint8 [%d,%d,3]" % (self.mg.height, self.mg.width))
        if frame.strides[2] != 1 or frame.strides[1] != 3:
            frame = np.ascontiguousarray(frame)
        _check(lib().bsx_live_set_input_frame(self.h, frame.ctypes.data, frame.strides[0]), self.mg.h, "bsx_live_set_input_frame")

    def get_output_mask(self, mask: np.ndarray) -> bool:
        if (not isinstance(mask, np.ndarray) or mask.dtype != np.uint8 or mask.shape != (self.mg.height, self.mg.width) or mask.strides[1] != 1
                or mask.strides[0] < self.mg.width or not mask.flags.writeable):
            # the C side writes height rows of width bytes (as process_host does)
            raise BsxError("mask must be a writable uint8 [%d,%d] array with unit inner stride" % (self.mg.height, self.mg.width))
        rc = lib().bsx_live_get_output_mask(self.h, mask.ctypes.data, mask.strides[0])
        if rc < 0:
            _check(rc, self.mg.h, "bsx_live_get_output_mask")
        return rc == 1

    def close(self):
        if getattr(self, "h", None):
            lib().bsx_live_delete(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def model_precompile(path: str, arch: str | None = None) -> str:
    """Host only: emit + compile (hipRTC) the kernel specialised to this model's graph into the code-object cache; returns the note."""
    buf = C.create_string_buffer(1 << 14)
    rc = lib().bsx_model_precompile(os.fsencode(path), arch.encode() if arch else None, buf, len(buf))
    if rc != 0:
        raise BsxError(buf.value.decode(errors="replace"))
    return buf.value.decode()


def model_kernel_source(path: str) -> str:
    """The generated HIP source of the specialised per-frame program ('' when the graph has none)."""
    buf = C.create_string_buffer(1 << 20)
    n = lib().bsx_model_kernel_source(os.fsencode(path), buf, len(buf))
    if n < 0:
        raise BsxError(buf.value.decode(errors="replace"))
    return buf.value.decode() if n > 0 else ""


def model_seg_source(path: str) -> str:
    """The generated HIP source of the graph-specialised segment kernels ('' when the plan has no segment kernels)."""
    buf = C.create_string_buffer(1 << 20)
    n = lib().bsx_model_seg_source(os.fsencode(path), buf, len(buf))
    if n < 0:
        raise BsxError(buf.value.decode(errors="replace"))
    return buf.value.decode() if n > 0 else ""


def gauss_coeff_words(ksize: int, shift: int = 0):
    """(c4 [4][9] u32, c2 [2][17] u32): the coefficient words gauss_blur_k multiplies with (host only, no GPU) — bsx_debug_gauss_coeffs"""
    c4 = np.zeros((4, 9), np.uint32)
    c2 = np.zeros((2, 17), np.uint32)
    _check(lib().bsx_debug_gauss_coeffs(int(ksize), int(shift), c4.ctypes.data_as(C.POINTER(C.c_uint32)), c2.ctypes.data_as(C.POINTER(C.c_uint32))), None,
           "bsx_debug_gauss_coeffs")
    return c4, c2


def model_describe(path: str) -> str:
    """Parse + plan a model on the host only (no GPU needed); raises BsxError with the loader's message."""
    buf = C.create_string_buffer(1 << 16)
    rc = lib().bsx_model_describe(os.fsencode(path), buf, len(buf))
    if rc != 0:
        raise BsxError(buf.value.decode(errors="replace"))
    return buf.value.decode()


# ---- reference-named entry points -----------------------------------------------------------------
def bs_tensorflow_version() -> str:
    return lib().bsx_version().decode()


def bs_maskgen_new(modelname, threads, width, height, ondebug=None, onprep=None, oninfer=None, onmask=None, caller_ctx=None):
    """Returns a context, or None on failure (after a message through ondebug/stderr) — like lib/libbackscrub.cc:161-259."""
    try:
        return MaskGen(modelname, width, height, n_streams=1, threads=threads, ondebug=ondebug, onprep=onprep, oninfer=oninfer,
                       onmask=onmask, caller_ctx=caller_ctx)
    except BsxError:
        return None


def bs_maskgen_delete(context):
    if context is not None:  # NULL-safe like :262
        context.close()


def bs_maskgen_process(context, frame: np.ndarray, mask: np.ndarray) -> bool:
    """frame: BGR uint8 [H,W,3]; mask: uint8 [H,W] written in place.  False on a null context / failure (:280)."""
    if context is None or getattr(context, "h", None) is None:
        return False
    try:
        context.process_host(frame, 0, mask)
        return True
    except BsxError:
        return False


def alpha_blend(srca, srcb, mask, context: MaskGen):
    """GPU alpha_blend (app/deepseg.cc:108-134): srca=background, srcb=frame, mask 255⇒srca. Torch cuda u8 tensors."""
    if srcb.dim() == 3:
        return context.composite(srca, srcb[None], mask[None])[0]
    return context.composite(srca, srcb, mask)
