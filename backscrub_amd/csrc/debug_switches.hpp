// debug_switches.hpp — the library's environment switches, in two classes (round 6).
//
// USER MODES are read with plain getenv() and are documented in include/bsx.h / README.md:
//   BSX_DEVICE (the C++ shim's GPU), BSX_F16_GEMM, BSX_ACT16 (opt-in reduced-precision modes), BSX_NO_UNIFORM_TILES (every mask tile on the general path),
//   BSX_KERNEL_CACHE / BSX_KERNEL_CACHE_OFF (where the graph-specialised kernels are cached).
//
// DEBUG SWITCHES — A/B timing knobs, alternate code paths kept for cross-checking, and experiments that make kernels SKIP WORK (results are then wrong on
// purpose) — go through BSX_DBG_ENV().  The default build compiles it to a null pointer: the switch names are not even in the binary, and setting them does nothing
// (tests/test_cabi.py: `strings libbsx.so`; tests/test_gpu_switch_variants.py runs the alternate paths against libbsx_dbg.so, built with -DBSX_DEBUG_SWITCHES).
#pragma once
#include <cstdlib>

#ifdef BSX_DEBUG_SWITCHES
#define BSX_DBG_ENV(name) getenv(name)
#else
#define BSX_DBG_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace bsx {
inline int dbg_env_int(const char* v, int dflt) { return v ? atoi(v) : dflt; }      // BSX_DBG_ENV("X") evaluated once: dbg_env_int(BSX_DBG_ENV("X"), default)
}
