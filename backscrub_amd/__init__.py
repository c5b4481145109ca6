"""backscrub_amd — MI355X-native implementation of backscrub's per-frame hot path.

The product is `libbsx.so` (HIP kernels + C++ host code behind the C ABI in
`include/bsx.h`).  This package is the thin Python host side used by tests and bench:
a ctypes binding plus mirrors of the reference's entry points
(`bs_maskgen_new / bs_maskgen_process / bs_maskgen_delete`, `alpha_blend`) with the same
names, argument meaning and error behaviour.  There is NO CPU fallback: if the HIP library
is missing or no GPU is visible, calls raise.
"""
from .api import (  # noqa: F401
    Background, BsxError, Live, MaskGen, alpha_blend, bs_maskgen_delete, bs_maskgen_new, bs_maskgen_process, bs_tensorflow_version, lib,
    lib_path, media_decode, model_describe,
)
