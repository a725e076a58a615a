// Shared host-side plumbing of libmistitch: error reporting, the context object,
// grow-only device workspaces.  No kernels here.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/mistitch.h"

namespace mistitch {

// ---- last-error string (one per calling thread) ---------------------------
inline char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define MS_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return ::mistitch::set_error(MISTITCH_ERR_HIP, "%s failed: %s (%s:%d)", #expr,      \
                                   hipGetErrorString(e__), __FILE__, __LINE__);           \
  } while (0)

#define MS_REQUIRE(cond, code, ...)                                   \
  do {                                                                \
    if (!(cond)) return ::mistitch::set_error((code), __VA_ARGS__);   \
  } while (0)

// ---- grow-only device buffer ------------------------------------------------
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t want) {
    if (want <= bytes) return MISTITCH_OK;
    if (ptr) {
      MS_HIP(hipFree(ptr));
      ptr = nullptr;
      bytes = 0;
    }
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) {
      ptr = nullptr;
      return set_error(MISTITCH_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    bytes = want;
    return MISTITCH_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
};

// ---- FFT plan (one per length and precision), tables live on the device ------
struct FftPlanDev {
  int n;
  int npass;
  int radix[6];
  const void* tw;             // n complex entries exp(-2 pi i k / n), precision of the plan
  const uint16_t* pos2freq;   // storage position -> frequency index (digit-reversed order)
  const uint16_t* freq2pos;   // inverse of pos2freq
  const uint16_t* mpos;       // storage position of the negated frequency
};

struct FftPlanHost {
  FftPlanDev dev{};
  void* blob = nullptr;  // single device allocation backing the tables
};

// ---- optional per-kernel timing (HIP events on the context's stream) ---------
struct Profiler {
  bool on = false;
  struct Rec {
    const char* name;
    hipEvent_t e0, e1;
  };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void begin(const char* name, hipStream_t st) {
    if (!on) return;
    Rec r{name, get(), get()};
    (void)hipEventRecord(r.e0, st);
    recs.push_back(r);
  }
  void end(hipStream_t st) {
    if (!on) return;
    (void)hipEventRecord(recs.back().e1, st);
  }
  void release() {
    for (auto& r : recs) {
      pool.push_back(r.e0);
      pool.push_back(r.e1);
    }
    recs.clear();
  }
  void destroy() {
    release();
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    pool.clear();
  }
};

}  // namespace mistitch

// Pins a loaded value in its register at this program point.  Placed after a
// group of independent loads it keeps the compiler from sinking each load down to
// its first use (which would serialise their latencies); the group is then issued
// back to back and waited for once.
#define MS_PIN(x) asm volatile("" : "+v"(x))

// Launch on the context's stream, timed when profiling is enabled.
#define MS_LAUNCH(ctx, name, kernel, grid, block, lds, ...)                     \
  do {                                                                          \
    (ctx)->prof.begin(name, (ctx)->stream);                                     \
    hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);   \
    (ctx)->prof.end((ctx)->stream);                                             \
  } while (0)

// The opaque handle of the C ABI.
struct mistitch_ctx {
  int device = 0;
  int n_cus = 256;  // compute units of the device
  hipStream_t stream = nullptr;
  int pair_chunk = 128;  // pairs processed per pipeline pass (bounds the FFT workspace)
  int64_t staging_limit = 4ll << 30;  // strips of all pairs of a call are staged at once up to this many bytes
  int upsample_factor = 100;  // subpixel grid: 1 / upsample_factor pixels
  int pixel_max = 65535;      // largest value of the tiles' pixel type: what a flatfield-corrected pixel is clipped to
  std::map<long long, mistitch::FftPlanHost> plans;  // key = n * 2 + (is_f64)
  mistitch::DevBuf desc, staging, spectrum, slabs, partial, peaks, cand, acc, misc, bins, flags, ups;
  std::vector<char> host_scratch;
  std::set<const void*> lds_enabled;  // kernels already granted large dynamic LDS
  std::map<long long, std::vector<char>> shapes;  // cached per-strip-shape launch configuration
  std::vector<void*> shape_allocs;                // device tables owned by those configurations
  std::map<int, std::vector<char>> gen_plans;     // chirp-z plans by transform length
  std::map<long long, void*> gen_tables;          // subpixel column tables of chirp-z strip shapes (freed with shape_allocs)
  std::map<int, void*> ups_grid_tables;           // subpixel node displacements and interpolation matrix by upsample factor (freed with shape_allocs)
  std::map<int, void*> hann_tables;               // symmetric Hann windows by length, float64 (freed with shape_allocs)
  mistitch::Profiler prof;
};
