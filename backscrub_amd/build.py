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
SOURCES = ["tflite_model.cpp", "plan.cpp", "kernels_nn.hip", "kernels_img.hip", "bsx_api.hip"]
HEADERS = ["tflite_model.hpp", "plan.hpp", "kernels.hpp", os.path.join("..", "..", "include", "bsx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
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
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
