// In-LDS power-of-two FFT engine for gfx950 wavefronts (64 lanes).
//
// Forward transforms are decimation-in-frequency, in place, natural order in,
// digit-reversed order out; inverse transforms are the adjoint passes run
// backwards (digit-reversed in, natural out).  A forward/inverse pair with
// pointwise work in between therefore needs no reordering pass at all: the
// pointwise stage looks frequencies up through the plan's pos2freq / mpos
// tables.  Each lane owns whole radix-R butterflies (R <= 16) in registers and
// LDS is touched once per pass.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.hpp"

namespace mistitch {

template <typename T>
using cx = typename std::conditional<std::is_same<T, float>::value, float2, double2>::type;

template <typename T>
__device__ __forceinline__ cx<T> mk(T x, T y) {
  cx<T> r;
  r.x = x;
  r.y = y;
  return r;
}
template <typename C>
__device__ __forceinline__ C cadd(C a, C b) {
  a.x += b.x;
  a.y += b.y;
  return a;
}
template <typename C>
__device__ __forceinline__ C csub(C a, C b) {
  a.x -= b.x;
  a.y -= b.y;
  return a;
}
template <typename C>
__device__ __forceinline__ C cmul(C a, C b) {
  // two products and two fused multiply-adds: the library is built with contraction off (the
  // pixel arithmetic of the fusion kernels must round every operation), so the fusing is spelled out
  // (complex64 keeps separate products and sums, which the compiler packs two to an instruction)
  C r;
  if constexpr (sizeof(a.x) == 8) {
    r.x = __builtin_fma(a.x, b.x, -(a.y * b.y));
    r.y = __builtin_fma(a.x, b.y, a.y * b.x);
  } else {
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
  }
  return r;
}
// a * conj(b)
template <typename C>
__device__ __forceinline__ C cmulc(C a, C b) {
  C r;
  if constexpr (sizeof(a.x) == 8) {
    r.x = __builtin_fma(a.x, b.x, a.y * b.y);
    r.y = __builtin_fma(a.y, b.x, -(a.x * b.y));
  } else {
    r.x = a.x * b.x + a.y * b.y;
    r.y = a.y * b.x - a.x * b.y;
  }
  return r;
}
template <typename C>
__device__ __forceinline__ C cconj(C a) {
  a.y = -a.y;
  return a;
}

// Multiply by exp(-+ 2 pi i K / 16) (minus for forward, plus for inverse).
template <typename T, int K, bool INV>
__device__ __forceinline__ cx<T> mul_w16(cx<T> a) {
  static_assert(K >= 0 && K < 8, "K out of range");
  constexpr T c1 = (T)0.92387953251128675613L;  // cos(pi/8)
  constexpr T s1 = (T)0.38268343236508977173L;  // sin(pi/8)
  constexpr T h = (T)0.70710678118654752440L;   // sqrt(1/2)
  if constexpr (K == 0) {
    return a;
  } else if constexpr (K == 4) {  // -+ i
    return INV ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x);
  } else if constexpr (K == 2) {  // (1 -+ i) / sqrt2
    return INV ? mk<T>((a.x - a.y) * h, (a.x + a.y) * h) : mk<T>((a.x + a.y) * h, (a.y - a.x) * h);
  } else if constexpr (K == 6) {  // (-1 -+ i) / sqrt2
    return INV ? mk<T>((-a.x - a.y) * h, (a.x - a.y) * h) : mk<T>((a.y - a.x) * h, (-a.x - a.y) * h);
  } else {
    constexpr T wr = (K == 1) ? c1 : (K == 3) ? s1 : (K == 5) ? -s1 : -c1;
    constexpr T wi0 = (K == 1) ? s1 : (K == 3) ? c1 : (K == 5) ? c1 : s1;  // sin(2 pi K / 16)
    constexpr T wi = INV ? wi0 : -wi0;
    return mk<T>(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
  }
}

// Radix-R DFT of v[0..R) in registers, natural order in and out.
template <typename T, int R, bool INV>
struct Dft;

template <typename T, bool INV>
struct Dft<T, 2, INV> {
  static __device__ __forceinline__ void run(cx<T> (&v)[2]) {
    cx<T> a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

template <typename T, int R, bool INV>
struct Dft {
  static __device__ __forceinline__ void run(cx<T> (&v)[R]) {
    constexpr int H = R / 2;
    cx<T> e[H], o[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      e[k] = v[2 * k];
      o[k] = v[2 * k + 1];
    }
    Dft<T, H, INV>::run(e);
    Dft<T, H, INV>::run(o);
    combine<0>(v, e, o);
  }
  template <int K>
  static __device__ __forceinline__ void combine(cx<T> (&v)[R], cx<T> (&e)[R / 2], cx<T> (&o)[R / 2]) {
    if constexpr (K < R / 2) {
      cx<T> t = mul_w16<T, K*(16 / R), INV>(o[K]);
      v[K] = cadd(e[K], t);
      v[K + R / 2] = csub(e[K], t);
      combine<K + 1>(v, e, o);
    }
  }
};

// Workgroup barrier that orders LDS traffic only.  The transforms communicate through LDS alone,
// and unlike __syncthreads() this does not wait for global loads a caller has in flight (the
// next block of input being fetched into registers during the arithmetic).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS addressing of element i of transform f.
//  Line: transforms are contiguous padded lines (pad one element per 256 B so
//        that the stride-R accesses of the late passes spread over all banks).
//  Col:  transforms are columns of a row-major [n][rowstride] matrix.
template <typename T>
struct LineAddr {
  int fstride;
  static constexpr int kPadShift = std::is_same<T, float>::value ? 5 : 4;
  __host__ __device__ __forceinline__ static int padded(int i) { return i + (i >> kPadShift); }
  __device__ __forceinline__ int operator()(int f, int i) const { return f * fstride + padded(i); }
  // work item w -> (transform f, butterfly g); per = butterflies per transform, a power of two
  __device__ __forceinline__ void split(int w, int nfft, int per, int& f, int& g) const {
    f = w >> (31 - __builtin_clz(per));
    g = w & (per - 1);
  }
};

template <typename T>
struct ColAddr {
  int rowstride;
  __device__ __forceinline__ int operator()(int f, int i) const { return i * rowstride + f; }
  // adjacent lanes take adjacent columns; nfft (columns) must be a power of two
  __device__ __forceinline__ void split(int w, int nfft, int per, int& f, int& g) const {
    g = w >> (31 - __builtin_clz(nfft));
    f = w & (nfft - 1);
  }
};

// Twiddle factors exp(-2 pi i k / N) come from two small LDS tables,
// w^k = hi[k >> 6] * lo[k & 63], and their powers from a short product tree: a
// pass then needs one lookup per butterfly instead of R - 1 global loads whose
// latency nothing in a barrier-separated pass could hide.
template <typename T>
struct Twiddles {
  const cx<T>* lo;  // w^k, k < 64
  const cx<T>* hi;  // w^(64 k), k < max(N / 64, 1)
  __device__ __forceinline__ cx<T> get(int k) const { return k < 64 ? lo[k] : cmul(hi[k >> 6], lo[k & 63]); }
  static __host__ __device__ __forceinline__ int entries(int n) { return 64 + (n >= 64 ? n / 64 : 1); }
};

// Fill the two tables (entries(plan.n) elements at dst); caller synchronises.
template <typename T>
__device__ __forceinline__ Twiddles<T> load_twiddles(cx<T>* dst, const FftPlanDev& plan, int tid, int nthr) {
  const cx<T>* tw = reinterpret_cast<const cx<T>*>(plan.tw);
  const int n_hi = plan.n >= 64 ? plan.n / 64 : 1;
  for (int k = tid; k < 64 + n_hi; k += nthr) {
    int src = k < 64 ? k : (k - 64) * 64;
    if (src >= plan.n) src = 0;
    dst[k] = tw[src];
  }
  return Twiddles<T>{dst, dst + 64};
}

// w[d] = base^d for d = 1 .. R-1 (squarings for even d, one extra product for odd d).
template <typename T, int R>
__device__ __forceinline__ void twiddle_powers(cx<T> base, cx<T> (&w)[R]) {
  w[1] = base;
#pragma unroll
  for (int d = 2; d < R; ++d) w[d] = (d & 1) ? cmul(w[d - 1], base) : cmul(w[d / 2], w[d / 2]);
}

// One radix-R pass over nfft transforms of length N whose current sub-block
// length is n (n = N on the first forward pass).
template <typename T, int R, bool INV, class Addr>
__device__ __forceinline__ void fft_pass(cx<T>* buf, const Addr& addr, int nfft, int N, int n,
                                         const Twiddles<T>& tw, int tid, int nthr) {
  // every length here is a power of two: shifts, not divisions
  const int per = N / R;
  const int sub = n / R;
  const int lsub = 31 - __builtin_clz(sub), ln = 31 - __builtin_clz(n);
  const int ltwstep = (31 - __builtin_clz(N)) - ln;
  for (int w = tid; w < nfft * per; w += nthr) {
    int f, g;
    addr.split(w, nfft, per, f, g);
    const int blk = g >> lsub;
    const int j = g & (sub - 1);
    const int base = (blk << ln) + j;
    cx<T> v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = buf[addr(f, base + m * sub)];
    if constexpr (!INV) {
      Dft<T, R, false>::run(v);
      if (sub > 1) {
        cx<T> pw[R];
        twiddle_powers<T, R>(tw.get(j << ltwstep), pw);
#pragma unroll
        for (int d = 1; d < R; ++d) v[d] = cmul(v[d], pw[d]);
      }
    } else {
      if (sub > 1) {
        cx<T> pw[R];
        twiddle_powers<T, R>(tw.get(j << ltwstep), pw);
#pragma unroll
        for (int d = 1; d < R; ++d) v[d] = cmulc(v[d], pw[d]);
      }
      Dft<T, R, true>::run(v);
    }
#pragma unroll
    for (int m = 0; m < R; ++m) buf[addr(f, base + m * sub)] = v[m];
  }
}

// MAXR: largest radix the calling kernel's plans may contain.  A kernel is given the registers
// of its widest butterfly whether or not a plan uses it, so kernels whose plans stop at
// radix 8 leave radix 16 out (complex128 radix-16 butterflies cost 154 VGPRs, radix-8 ones 90).
template <typename T, bool INV, int MAXR, class Addr>
__device__ __forceinline__ void fft_dispatch(int radix, cx<T>* buf, const Addr& addr, int nfft, int N,
                                             int n, const Twiddles<T>& tw, int tid, int nthr) {
  switch (radix) {
    case 16:
      if constexpr (MAXR >= 16) fft_pass<T, 16, INV>(buf, addr, nfft, N, n, tw, tid, nthr);
      break;
    case 8: fft_pass<T, 8, INV>(buf, addr, nfft, N, n, tw, tid, nthr); break;
    case 4: fft_pass<T, 4, INV>(buf, addr, nfft, N, n, tw, tid, nthr); break;
    default: fft_pass<T, 2, INV>(buf, addr, nfft, N, n, tw, tid, nthr); break;
  }
}

// Forward transform of nfft sequences held in LDS.  Caller must have
// synchronised after filling buf and the twiddle tables; returns with the block
// synchronised.
template <typename T, int MAXR = 16, class Addr>
__device__ __forceinline__ void fft_forward(cx<T>* buf, const Addr& addr, int nfft, const FftPlanDev& plan,
                                            const Twiddles<T>& tw, int tid, int nthr) {
  int n = plan.n;
  for (int i = 0; i < plan.npass; ++i) {
    fft_dispatch<T, false, MAXR>(plan.radix[i], buf, addr, nfft, plan.n, n, tw, tid, nthr);
    n /= plan.radix[i];
    lds_barrier();
  }
}

// Unnormalised inverse transform (digit-reversed in, natural out).
template <typename T, int MAXR = 16, class Addr>
__device__ __forceinline__ void fft_inverse(cx<T>* buf, const Addr& addr, int nfft, const FftPlanDev& plan,
                                            const Twiddles<T>& tw, int tid, int nthr) {
  int n = 1;
  for (int i = plan.npass - 1; i >= 0; --i) {
    n *= plan.radix[i];
    fft_dispatch<T, true, MAXR>(plan.radix[i], buf, addr, nfft, plan.n, n, tw, tid, nthr);
    lds_barrier();
  }
}

}  // namespace mistitch
