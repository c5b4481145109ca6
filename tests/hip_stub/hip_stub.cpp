// hip_stub.cpp — TEST INFRASTRUCTURE (tests/test_device_order.py), never part of the product.
//
// An LD_PRELOAD interposer for the HIP runtime entry points libbsx.so imports, for a box WITHOUT a GPU: "device memory" is host memory, launches do nothing, and every
// call is logged with the calling thread's CURRENT device.  It pretends to have BSX_STUB_NDEV (default 2) devices, so the code path `device != 0` — which no
// single-GPU test box can execute — runs its real host code (plan, hipRTC cache load, per-device attributes, hipGraph capture, every entry point's device guard)
// and the test can assert that no device-affine HIP call is made while another device is current, that streams / events / allocations are only used on the device
// they were created on, and that the caller's device is restored.  Results of the "kernels" are meaningless; order and device affinity are what is checked.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
thread_local int t_dev = 0;
std::mutex g_mu;
std::map<const void*, int> g_owner;       // stream / event / allocation / module / graph → device it was created on
FILE* g_log = nullptr;
int g_capture = 0;

int ndev() { const char* e = getenv("BSX_STUB_NDEV"); return e ? atoi(e) : 2; }
void logf(const char* api, const char* kind, const char* note = "") {
  std::lock_guard<std::mutex> l(g_mu);
  if (!g_log) { const char* p = getenv("BSX_STUB_LOG"); g_log = p ? fopen(p, "a") : nullptr; if (!g_log) return; }
  fprintf(g_log, "%s %s %d %s\n", kind, api, t_dev, note);
  fflush(g_log);
}
void own(const void* p) { std::lock_guard<std::mutex> l(g_mu); g_owner[p] = t_dev; }
void disown(const void* p) { std::lock_guard<std::mutex> l(g_mu); g_owner.erase(p); }
// a handle used while a device other than its creator's is current → "MISMATCH" line
void use(const char* api, const void* p) {
  if (!p) return;
  int o = -1;
  { std::lock_guard<std::mutex> l(g_mu); auto it = g_owner.find(p); if (it != g_owner.end()) o = it->second; }
  if (o >= 0 && o != t_dev) { char b[64]; snprintf(b, sizeof b, "owner=%d", o); logf(api, "MISMATCH", b); }
}
void* handle() { return calloc(1, 64); }
}  // namespace

extern "C" {
int bsx_stub_current_device() { return t_dev; }
void bsx_stub_set_device(int d) { t_dev = d; }

hipError_t hipGetDeviceCount(int* n) { *n = ndev(); logf("hipGetDeviceCount", "neutral"); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= ndev()) return hipErrorInvalidDevice; t_dev = d; logf("hipSetDevice", "neutral"); return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int dev) {
  memset(p, 0, sizeof *p);
  snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  char b[32]; snprintf(b, sizeof b, "queried=%d", dev);
  logf("hipGetDeviceProperties", "neutral", b);
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { logf("hipDeviceSynchronize", "affine"); return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); own(*p); logf("hipMalloc", "affine"); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { use("hipFree", p); disown(p); free(p); logf("hipFree", "affine"); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); logf("hipHostMalloc", "affine"); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); logf("hipHostFree", "affine"); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { use("hipMemcpy", d); use("hipMemcpy", s); memmove(d, s, n); logf("hipMemcpy", "affine"); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  use("hipMemcpyAsync", d); use("hipMemcpyAsync", s); use("hipMemcpyAsync", st);
  if (!g_capture) memmove(d, s, n);
  logf("hipMemcpyAsync", "affine");
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st) {
  use("hipMemcpy2DAsync", d); use("hipMemcpy2DAsync", s); use("hipMemcpy2DAsync", st);
  if (!g_capture) for (size_t y = 0; y < h; y++) memmove((char*)d + y * dp, (const char*)s + y * sp, w);
  logf("hipMemcpy2DAsync", "affine");
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { use("hipMemset", d); memset(d, v, n); logf("hipMemset", "affine"); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { use("hipMemsetAsync", d); use("hipMemsetAsync", st); memset(d, v, n); logf("hipMemsetAsync", "affine"); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)handle(); own(*s); logf("hipStreamCreateWithFlags", "affine"); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)handle(); own(*s); logf("hipStreamCreateWithPriority", "affine"); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; logf("hipDeviceGetStreamPriorityRange", "affine"); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { use("hipStreamDestroy", s); disown(s); free(s); logf("hipStreamDestroy", "affine"); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { use("hipStreamSynchronize", s); logf("hipStreamSynchronize", "affine"); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { use("hipStreamWaitEvent", s); use("hipStreamWaitEvent", e); logf("hipStreamWaitEvent", "affine"); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)handle(); own(*e); logf("hipEventCreate", "affine"); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)handle(); own(*e); logf("hipEventCreateWithFlags", "affine"); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { use("hipEventDestroy", e); disown(e); free(e); logf("hipEventDestroy", "affine"); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { use("hipEventRecord", e); use("hipEventRecord", s); logf("hipEventRecord", "affine"); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { use("hipEventQuery", e); logf("hipEventQuery", "affine"); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { use("hipEventElapsedTime", a); use("hipEventElapsedTime", b); *ms = 1.0f; logf("hipEventElapsedTime", "affine"); return hipSuccess; }

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { logf("hipFuncSetAttribute", "affine"); return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t s) { use("launch", s); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) { *g = dim3(1, 1, 1); *b = dim3(1, 1, 1); *sh = 0; *s = nullptr; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t s) { use("hipLaunchKernel", s); logf("hipLaunchKernel", "affine"); return hipSuccess; }
hipError_t hipModuleLoadData(hipModule_t* m, const void*) { *m = (hipModule_t)handle(); own(*m); logf("hipModuleLoadData", "affine"); return hipSuccess; }
hipError_t hipModuleUnload(hipModule_t m) { use("hipModuleUnload", m); disown(m); free(m); logf("hipModuleUnload", "affine"); return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char*) { use("hipModuleGetFunction", m); *f = (hipFunction_t)handle(); own(*f); logf("hipModuleGetFunction", "affine"); return hipSuccess; }
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t s, void**, void**) {
  use("hipModuleLaunchKernel", f); use("hipModuleLaunchKernel", s); logf("hipModuleLaunchKernel", "affine"); return hipSuccess;
}

hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { use("hipStreamBeginCapture", s); g_capture++; logf("hipStreamBeginCapture", "affine"); return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) { use("hipStreamEndCapture", s); g_capture--; *g = (hipGraph_t)handle(); own(*g); logf("hipStreamEndCapture", "affine"); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) { use("hipGraphInstantiate", g); *e = (hipGraphExec_t)handle(); own(*e); logf("hipGraphInstantiate", "affine"); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { use("hipGraphDestroy", g); disown(g); free(g); logf("hipGraphDestroy", "affine"); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { use("hipGraphExecDestroy", e); disown(e); free(e); logf("hipGraphExecDestroy", "affine"); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) { use("hipGraphLaunch", e); use("hipGraphLaunch", s); logf("hipGraphLaunch", "affine"); return hipSuccess; }
}
