"""Build libbsx.so (the HIP/C++ product library) in-tree for gfx950.

    python -m backscrub_amd.build [--force]

One hipcc invocation per translation unit (objects cached under backscrub_amd/csrc/build/),
then a shared-library link.  The .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libbsx.so")
# the same sources with -DBSX_DEBUG_SWITCHES (csrc/debug_switches.hpp): the A/B knobs, alternate code paths and work-skipping experiments the default build leaves
# out.  Test infrastructure — tests/test_gpu_switch_variants.py and the tools/ experiments load it through BSX_LIBRARY; nothing ships or measures with it.
OBJ_DBG = os.path.join(CSRC, "build_dbg")
LIB_DBG = os.path.join(HERE, "libbsx_dbg.so")
SOURCES = ["tflite_model.cpp", "plan.cpp", "gen_mid.cpp", "gen_seg.cpp", "rtc.cpp", "media.cpp", "jpeg.cpp", "live.cpp", "kernels_nn.hip", "kernels_img.hip", "kernels_frame.hip", "kernels_seg.hip", "bsx_api.hip"]
HEADERS = ["mid_prelude.hip", "debug_switches.hpp", "gen_mid.hpp", "gen_seg.hpp", "rtc.hpp", "media.hpp", "tflite_model.hpp", "plan.hpp", "kernels.hpp", "frame_program.hpp", "segments.hpp", "mfma_tile.hpp", "roctx_ranges.hpp", os.path.join("..", "..", "include", "bsx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]      # only the BSX_API entry points of include/bsx.h are exported


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def strip_line_comments(text):
    """`// ...` comments removed (not inside string literals), line structure kept — hipRTC's diagnostics keep their line numbers, the embedded text loses a third of
    its bytes and every switch name the comments mention."""
    out = []
    for line in text.split("\n"):
        i, in_str, res = 0, False, None
        while i < len(line):
            ch = line[i]
            if in_str:
                if ch == "\\":
                    i += 1
                elif ch == '"':
                    in_str = False
            elif ch == '"':
                in_str = True
            elif ch == "/" and line[i:i + 2] == "//":
                res = line[:i].rstrip()
                break
            i += 1
        out.append(line if res is None else res)
    return "\n".join(out)


def embed_prelude():
    """csrc/mid_prelude.hip → csrc/build/mid_prelude_str.inc: the device templates (comments stripped) as a C++ raw string literal that gen_mid.cpp includes
    (the specialised kernels are compiled from it by hipRTC when a context is created)."""
    src = os.path.join(CSRC, "mid_prelude.hip")
    text = strip_line_comments(open(src).read())
    assert ')BSXRTC"' not in text
    body = 'R"BSXRTC(' + text + ')BSXRTC"\n'
    for d in (OBJ, OBJ_DBG):
        dst = os.path.join(d, "mid_prelude_str.inc")
        if not os.path.exists(dst) or open(dst).read() != body:
            with open(dst, "w") as f:
                f.write(body)


SEG_RTC_PREAMBLE = """// hipRTC translation unit of the graph-specialised segment kernels (gen_seg.cpp puts its macros in front and the descriptors at the marker inside namespace segrtc)
#define BSX_DBG_ENV(name) ((const char*)0)
namespace bsx { enum Activation : int { kActNone = 0, kActRelu = 1, kActRelu6 = 3, kActHswish = 100, kActSigmoid = 101 }; }
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
"""


def embed_seg_source():
    """csrc/segments.hpp + mfma_tile.hpp + kernels_seg.hip → <objdir>/seg_rtc_src.inc: ONE flattened, comment-stripped text as a C++ raw string literal (gen_seg.cpp) —
    what hipRTC compiles, with the loaded graph's descriptors as constants, when a context is created.  The files guard their host-only parts with __HIPCC_RTC__."""
    parts = [SEG_RTC_PREAMBLE]
    for f in ("segments.hpp", "mfma_tile.hpp", "kernels_seg.hip"):
        t = strip_line_comments(open(os.path.join(CSRC, f)).read())
        t = "\n".join(l for l in t.split("\n") if l.strip() != "#pragma once")
        parts.append(t)
    text = "\n".join(parts)
    # every macro of the embedded text gets its own prefix: the release library's strings carry no BSX_ name but the documented user modes (tests/test_cabi.py)
    text = text.replace("BSX_", "BSXS_")
    assert ')BSXSEG"' not in text and "BSXS_SEG_CONSTANTS" in text
    # string literals are limited in length by some compilers: cut into adjacent raw strings (concatenated by the compiler)
    chunks = [text[i:i + 8000] for i in range(0, len(text), 8000)]
    body = "\n".join('R"BSXSEG(' + c + ')BSXSEG"' for c in chunks) + "\n"
    for d in (OBJ, OBJ_DBG):
        dst = os.path.join(d, "seg_rtc_src.inc")
        if not os.path.exists(dst) or open(dst).read() != body:
            with open(dst, "w") as f:
                f.write(body)


def _build_one(objdir, lib, extra_flags, force, verbose):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(objdir, "mid_prelude_str.inc"), os.path.join(objdir, "seg_rtc_src.inc")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + extra_flags + ["-I", objdir] + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("==== hipcc failed for %s ====\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("libbsx build failed")
    if force or procs or _stale(lib, objs + [os.path.join(CSRC, "libbsx.map")]):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,--version-script=" + os.path.join(CSRC, "libbsx.map"), "-o", lib] + objs + ["-lz", "-lpthread", "-lhiprtc", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return bool(procs)


def build(force=False, verbose=False, debug_lib=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(OBJ_DBG, exist_ok=True)
    embed_prelude()
    embed_seg_source()
    rebuilt = _build_one(OBJ, LIB, [], force, verbose)
    if debug_lib:
        _build_one(OBJ_DBG, LIB_DBG, ["-DBSX_DEBUG_SWITCHES"], force, verbose)
    build_shim_demo(force or rebuilt, verbose)
    return LIB


DEMO = os.path.join(HERE, "bsx_demo")


def build_shim_demo(force=False, verbose=False):
    """C++ drop-in shim (bs_maskgen_* over the C ABI) + the demo application, compiled with plain g++
    against tests/cv_stub (OpenCV is not installed in this image)."""
    root = os.path.dirname(HERE)
    shim = os.path.join(CSRC, "bs_maskgen_shim.cpp")
    demo_src = os.path.join(root, "tools", "bsx_demo.cpp")
    deps = [shim, demo_src, os.path.join(root, "include", "bsx.h"), os.path.join(root, "include", "bs_maskgen.h"), LIB]
    if not (force or _stale(DEMO, deps)):
        return DEMO
    inc = ["-I", os.path.join(root, "include"), "-I", os.path.join(root, "tests", "cv_stub")]
    obj = os.path.join(OBJ, "bs_maskgen_shim.o")
    cmds = [["g++", "-std=c++17", "-O2", "-fPIC"] + inc + ["-c", shim, "-o", obj],
            ["g++", "-std=c++17", "-O2"] + inc + [demo_src, obj, "-L", HERE, "-lbsx", "-Wl,-rpath,$ORIGIN", "-o", DEMO]]
    for c in cmds:
        if verbose:
            print(" ".join(c))
        subprocess.check_call(c)
    return DEMO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
