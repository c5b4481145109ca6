// rtc.hpp — run-time compilation of graph-specialised kernels (hipRTC) with an on-disk code-object cache.
//
// The reference builds its execution plan when the context is created (InterpreterBuilder / AllocateTensors,
// /root/reference/lib/libbackscrub.cc:205-217); here that step also emits and compiles a kernel specialised to the loaded
// graph (gen_mid.cpp).  Compilation needs no GPU (the target architecture is given explicitly), so `bsx_model_precompile`
// can fill the cache on a build machine; a context on the GPU box then only loads the code object.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace bsx {

struct RtcKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
};

// Directory of the code-object cache: $BSX_KERNEL_CACHE, else <directory of libbsx.so>/kcache (created on demand; falls back to
// /tmp/bsx_kcache_<uid> when that is not writable).
std::string rtc_cache_dir();

// Compile `source` for `arch` ("gfx950"; features after ':' are ignored) or fetch the code object of an identical earlier
// compilation from the cache.  Returns false with the compiler log in `log`.  *cached tells where the code came from.
bool rtc_build(const std::string& source, const std::string& arch, std::vector<char>* code, std::string* log, bool* cached = nullptr);

// Bytes of scratch (private segment: register spills) per lane of `kernel` in a code object, read from its kernel descriptor (`<kernel>.kd`, bytes 4-7) — no GPU
// needed; -1 if the ELF cannot be read.
long code_object_scratch_bytes(const std::vector<char>& code, const char* kernel);

// Load a code object on the CURRENT device and resolve `kernel`.
hipError_t rtc_load(const std::vector<char>& code, const char* kernel, RtcKernel* out);
void rtc_unload(RtcKernel* k);
// one more kernel of a module already loaded by rtc_load (the module stays owned by the RtcKernel it was loaded into)
hipError_t rtc_function(const RtcKernel& loaded, const char* kernel, hipFunction_t* fn);

}  // namespace bsx
