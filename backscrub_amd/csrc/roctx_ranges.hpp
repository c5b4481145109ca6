// ROCTx ranges around the three stages the reference brackets with its callbacks (onprep / oninfer / onmask: lib/libbackscrub.cc:303,311,363) and the
// application times with timinginfo_t (app/deepseg.cc:137-156,701-720): `rocprofv3 --marker-trace --kernel-trace` then shows "bsx:prep", "bsx:network",
// "bsx:decode", "bsx:mask", "bsx:mask+blend" ranges next to the kernels they enqueue (tracing row).
//
// The marker library is resolved at run time (dlopen of librocprofiler-sdk-roctx, then the older libroctx64): libbsx.so has no link-time dependency on a
// profiler, a box without either library runs unmarked, and BSX_NO_ROCTX=1 skips the lookup.  Without a tool attached a push / pop pair is two indirect calls
// into a library that finds no registered client — tens of nanoseconds against launches of tens of microseconds.
#pragma once
#include "debug_switches.hpp"
#include <dlfcn.h>

#include <cstdlib>

namespace bsx_roctx {
using push_fn = int (*)(const char*);
using pop_fn = int (*)();
struct Api {
  push_fn push = nullptr;
  pop_fn pop = nullptr;
  Api() {
    if (BSX_DBG_ENV("BSX_NO_ROCTX")) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
      if (!h) continue;
      push = (push_fn)dlsym(h, "roctxRangePushA");
      pop = (pop_fn)dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
inline const Api& api() { static const Api a; return a; }
// host-side range (the enqueue of a stage's launches); nests
struct Range {
  bool on;
  explicit Range(const char* name) : on(api().push != nullptr) { if (on) api().push(name); }
  ~Range() { if (on) api().pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};
inline bool available() { return api().push != nullptr; }
}  // namespace bsx_roctx
