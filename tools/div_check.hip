// Exhaustive check of shortened float32 division chains against the IEEE-exact
// quotient, for the operands the flatfield correction meets: numerators that are
// integers 0..65535 and every denominator mantissa (one binade: scaling the
// denominator by a power of two scales every intermediate exactly, as long as
// nothing overflows or goes denormal).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/div_check.hip -o tools/div_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

constexpr int kChains = 5;

template <int C>
__device__ __forceinline__ float chain(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  if (C == 0 || C == 2 || C == 4) {  // one Newton step on the reciprocal
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
  }
  float q = n * r;
  float t = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(t, r, q);
  if (C == 0 || C == 1) {  // second residual correction
    t = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(t, r, q);
  }
  if (C == 4) {  // truncation-only use: nothing more
  }
  return q;
}

// chain 0: Newton + 2 corrections (the full expansion without scaling)
// chain 1: 2 corrections, raw reciprocal
// chain 2: Newton + 1 correction
// chain 3: 1 correction, raw reciprocal
// chain 4: as 2, compared after truncation to an integer only
__global__ __launch_bounds__(256) void k_check(unsigned mant0, unsigned long long* mismatches, unsigned* first_bad) {
  const unsigned mant = mant0 + blockIdx.x * 256 + threadIdx.x;
  const float d = __uint_as_float(0x3f800000u | (mant & 0x7fffffu));
  unsigned long long bad[kChains] = {0, 0, 0, 0, 0};
  for (unsigned v = 0; v < 65536; ++v) {
    const float n = (float)v;
    const float want = __fdiv_rn(n, d);
    const float c0 = chain<0>(n, d), c1 = chain<1>(n, d), c2 = chain<2>(n, d), c3 = chain<3>(n, d);
    bad[0] += c0 != want;
    bad[1] += c1 != want;
    bad[2] += c2 != want;
    bad[3] += c3 != want;
    bad[4] += (unsigned)c2 != (unsigned)want;
    if (c0 != want && atomicCAS(first_bad, 0u, 1u) == 0u) {
      first_bad[1] = mant;
      first_bad[2] = v;
    }
  }
  for (int c = 0; c < kChains; ++c)
    if (bad[c]) atomicAdd(mismatches + c, bad[c]);
}

int main(int argc, char** argv) {
  // a full binade is 2^23 mantissas; the optional argument limits it (in units of 2^16) for a quick run
  const unsigned chunks = argc > 1 ? (unsigned)atoi(argv[1]) : 128;
  unsigned long long* mism;
  unsigned* first;
  CK(hipMalloc(&mism, kChains * sizeof(unsigned long long)));
  CK(hipMalloc(&first, 4 * sizeof(unsigned)));
  CK(hipMemset(mism, 0, kChains * sizeof(unsigned long long)));
  CK(hipMemset(first, 0, 4 * sizeof(unsigned)));
  for (unsigned c = 0; c < chunks; ++c) {
    k_check<<<256, 256>>>(c * 65536u, mism, first);
    CK(hipDeviceSynchronize());
    if (c % 16 == 15) {
      printf("chunk %u of %u done\n", c + 1, chunks);
      fflush(stdout);
    }
  }
  unsigned long long h[kChains];
  unsigned hf[4];
  CK(hipMemcpy(h, mism, sizeof(h), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost));
  const double total = (double)chunks * 65536.0 * 65536.0;
  const char* names[kChains] = {"newton+2corr", "raw+2corr", "newton+1corr", "raw+1corr", "newton+1corr, truncated"};
  for (int c = 0; c < kChains; ++c) printf("%-26s mismatches %llu of %.3g\n", names[c], h[c], total);
  if (hf[0]) printf("first newton+2corr mismatch: mantissa %u numerator %u\n", hf[1], hf[2]);
  return 0;
}
