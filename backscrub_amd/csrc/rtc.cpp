// rtc.cpp — hipRTC compilation + on-disk cache of the graph-specialised kernels (see rtc.hpp).
#include "rtc.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>

namespace bsx {
namespace {

// 128-bit FNV-1a style digest as two independent 64-bit lanes: cache keys only, not cryptography
std::string digest(const std::string& s) {
  unsigned long long a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
  for (unsigned char ch : s) { a = (a ^ ch) * 0x100000001b3ull; b = (b ^ (ch + 0x9e)) * 0x100000001b3ull; b ^= b >> 29; }
  char buf[40];
  snprintf(buf, sizeof buf, "%016llx%016llx", a, b);
  return buf;
}

bool writable_dir(const std::string& d) {
  struct stat st;
  if (stat(d.c_str(), &st) != 0 && mkdir(d.c_str(), 0755) != 0) return false;
  return access(d.c_str(), W_OK | X_OK) == 0;
}

// A cached code object is LOADED AND RUN on the GPU and its name is computable by anyone (the generated source is deterministic), so a directory the library picks
// by itself must be one nobody else can write to: a real directory (not a symlink), owned by this user, no group / world write bit.  Created 0700 when missing.
bool private_dir(const std::string& d) {
  struct stat st;
  if (lstat(d.c_str(), &st) != 0) {
    if (mkdir(d.c_str(), 0700) != 0 || lstat(d.c_str(), &st) != 0) return false;
  }
  if (!S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 022) != 0) return false;
  return access(d.c_str(), W_OK | X_OK) == 0;
}

std::string lib_dir() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&rtc_cache_dir), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    if (k != std::string::npos) return p.substr(0, k);
  }
  return ".";
}

std::mutex g_mu;

}  // namespace

// BSX_KERNEL_CACHE (the user's explicit choice) → <library directory>/kcache → $XDG_CACHE_HOME/bsx_kcache → ~/.cache/bsx_kcache → /tmp/bsx_kcache_<uid>; every
// directory but the first must pass private_dir().  "" = no usable directory: the cache is off and every context compiles its kernel.
std::string rtc_cache_dir() {
  if (const char* e = getenv("BSX_KERNEL_CACHE")) { if (*e && writable_dir(e)) return e; }
  const std::string d = lib_dir() + "/kcache";
  if (private_dir(d)) return d;
  if (const char* x = getenv("XDG_CACHE_HOME")) { if (*x == '/' && private_dir(std::string(x) + "/bsx_kcache")) return std::string(x) + "/bsx_kcache"; }
  if (const char* h = getenv("HOME")) {
    if (*h == '/') { const std::string c = std::string(h) + "/.cache"; (void)mkdir(c.c_str(), 0700); if (private_dir(c + "/bsx_kcache")) return c + "/bsx_kcache"; }
  }
  const std::string t = "/tmp/bsx_kcache_" + std::to_string((long)getuid());
  return private_dir(t) ? t : std::string();
}

bool rtc_build(const std::string& source, const std::string& arch_in, std::vector<char>* code, std::string* log, bool* cached) {
  std::string arch = arch_in.substr(0, arch_in.find(':'));
  if (arch.empty()) arch = "gfx950";
  int rtc_major = 0, rtc_minor = 0;
  hiprtcVersion(&rtc_major, &rtc_minor);
  const std::string opts_key = arch + "|O3|no-contract|" + std::to_string(rtc_major) + "." + std::to_string(rtc_minor);
  const std::string dir = rtc_cache_dir();
  const bool cache_on = !dir.empty() && !getenv("BSX_KERNEL_CACHE_OFF");
  const std::string path = dir + "/" + digest(opts_key + "\n" + source) + ".hsaco";
  if (cached) *cached = false;
  std::lock_guard<std::mutex> lock(g_mu);
  if (cache_on) {
    std::ifstream f(path, std::ios::binary);
    if (f) {
      code->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      if (code->size() > 64 && memcmp(code->data(), "\x7f" "ELF", 4) == 0) { if (cached) *cached = true; return true; }
    }
  }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, source.c_str(), "bsx_specialised.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { if (log) *log = "hiprtcCreateProgram failed"; return false; }
  const std::string a = "--offload-arch=" + arch;
  const char* opts[] = {a.c_str(), "-O3", "-ffp-contract=off", "-std=c++17"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  if (log && ls > 1) { log->assign(ls, '\0'); hiprtcGetProgramLog(prog, &(*log)[0]); }
  if (r != HIPRTC_SUCCESS) { if (log && log->empty()) *log = hiprtcGetErrorString(r); hiprtcDestroyProgram(&prog); return false; }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->assign(cs, '\0');
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  if (cache_on) {                                                 // write to a temporary name, then rename: concurrent contexts / ranks never see half a file
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    std::ofstream o(tmp, std::ios::binary);
    if (o) {
      o.write(code->data(), (std::streamsize)code->size());
      o.close();
      if (!o || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
    }
  }
  return true;
}

// ELF64 little-endian (the only code object form hipRTC emits for amdgcn): section headers -> the symbol tables -> `<kernel>.kd` -> the section that holds its
// address -> the 64-byte kernel descriptor, whose bytes 4-7 are PRIVATE_SEGMENT_FIXED_SIZE.  Every offset is bounds-checked: the cache directory is user-writable.
long code_object_scratch_bytes(const std::vector<char>& code, const char* kernel) {
  const size_t N = code.size();
  const unsigned char* p = reinterpret_cast<const unsigned char*>(code.data());
  auto u16 = [&](size_t o) -> uint64_t { return o + 2 <= N ? (uint64_t)p[o] | ((uint64_t)p[o + 1] << 8) : 0; };
  auto u32 = [&](size_t o) -> uint64_t { return o + 4 <= N ? u16(o) | (u16(o + 2) << 16) : 0; };
  auto u64 = [&](size_t o) -> uint64_t { return o + 8 <= N ? u32(o) | (u32(o + 4) << 32) : 0; };
  if (N < 64 || memcmp(p, "\x7f" "ELF", 4) != 0 || p[4] != 2 || p[5] != 1) return -1;
  const uint64_t shoff = u64(0x28), shentsize = u16(0x3A), shnum = u16(0x3C);
  if (!shoff || shentsize < 64 || shnum == 0 || shoff > N || shnum * shentsize > N - shoff) return -1;
  const std::string want = std::string(kernel) + ".kd";
  auto sh = [&](uint64_t i, size_t field) { return shoff + i * shentsize + field; };
  for (uint64_t i = 0; i < shnum; i++) {
    const uint64_t type = u32(sh(i, 4));
    if (type != 2 && type != 11) continue;                          // SHT_SYMTAB / SHT_DYNSYM
    const uint64_t off = u64(sh(i, 0x18)), size = u64(sh(i, 0x20)), link = u32(sh(i, 0x28)), ent = u64(sh(i, 0x38));
    if (ent < 24 || link >= shnum || off > N || size > N - off) continue;
    const uint64_t stroff = u64(sh(link, 0x18)), strsize = u64(sh(link, 0x20));
    if (stroff > N || strsize > N - stroff) continue;
    for (uint64_t s = 0; s + ent <= size; s += ent) {
      const uint64_t name = u32(off + s), value = u64(off + s + 8);
      if (name >= strsize || strsize - name <= want.size()) continue;
      if (memcmp(p + stroff + name, want.c_str(), want.size() + 1) != 0) continue;
      for (uint64_t j = 0; j < shnum; j++) {                        // the section that holds the descriptor
        const uint64_t addr = u64(sh(j, 0x10)), so = u64(sh(j, 0x18)), ss = u64(sh(j, 0x20)), st = u32(sh(j, 4));
        if (st == 8 || value < addr || value - addr + 64 > ss) continue;      // (SHT_NOBITS has no bytes)
        const uint64_t fo = so + (value - addr);
        if (fo + 8 > N) return -1;
        return (long)u32(fo + 4);
      }
      return -1;
    }
  }
  return -1;
}

hipError_t rtc_load(const std::vector<char>& code, const char* kernel, RtcKernel* out) {
  hipError_t e = hipModuleLoadData(&out->mod, code.data());
  if (e != hipSuccess) return e;
  e = hipModuleGetFunction(&out->fn, out->mod, kernel);
  if (e != hipSuccess) { (void)hipModuleUnload(out->mod); out->mod = nullptr; out->fn = nullptr; }
  return e;
}

hipError_t rtc_function(const RtcKernel& loaded, const char* kernel, hipFunction_t* fn) {
  if (!loaded.mod) return hipErrorInvalidValue;
  return hipModuleGetFunction(fn, loaded.mod, kernel);
}

void rtc_unload(RtcKernel* k) {
  if (k && k->mod) { (void)hipModuleUnload(k->mod); k->mod = nullptr; k->fn = nullptr; }
}

}  // namespace bsx
