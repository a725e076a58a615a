// In-LDS FFT engine for gfx950 wavefronts (64 lanes): power-of-two lengths, and (at the end of the file)
// lengths whose factors are all at most 13.
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

// cos and sin of 2 pi k / R for the odd prime radices, k = 0 .. (R - 1) / 2 (the other half by symmetry).
template <int R>
struct OddRoots;
template <>
struct OddRoots<3> {
  static constexpr double c[2] = {1.0, -0.5};
  static constexpr double s[2] = {0.0, 0.86602540378443864676};
};
template <>
struct OddRoots<5> {
  static constexpr double c[3] = {1.0, 0.30901699437494742410, -0.80901699437494742410};
  static constexpr double s[3] = {0.0, 0.95105651629515357211, 0.58778525229247312917};
};
template <>
struct OddRoots<7> {
  static constexpr double c[4] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624};
  static constexpr double s[4] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048};
};
template <>
struct OddRoots<11> {
  static constexpr double c[6] = {1.0, 0.84125353283118116886, 0.41541501300188642553, -0.14231483827328514044,
                                  -0.65486073394528506406, -0.95949297361449738989};
  static constexpr double s[6] = {0.0, 0.54064081745559758210, 0.90963199535451837141, 0.98982144188093273238,
                                  0.75574957435425828377, 0.28173255684142969771};
};
template <>
struct OddRoots<13> {
  static constexpr double c[7] = {1.0, 0.88545602565320989590, 0.56806474673115580251, 0.12053668025532305335,
                                  -0.35460488704253562597, -0.74851074817110109863, -0.97094181742605202716};
  static constexpr double s[7] = {0.0, 0.46472317204376854566, 0.82298386589365639458, 0.99270887409805399280,
                                  0.93501624268541482344, 0.66312265824079520238, 0.23931566428755776695};
};
template <int R>
constexpr double odd_root_cos(int k) {
  return k <= R / 2 ? OddRoots<R>::c[k] : OddRoots<R>::c[R - k];
}
template <int R>
constexpr double odd_root_sin(int k) {
  return k <= R / 2 ? OddRoots<R>::s[k] : -OddRoots<R>::s[R - k];
}

// DFT of odd prime length R (complex128) by the symmetry of its roots: with a_k = v_k + v_{R-k} and
// b_k = v_k - v_{R-k}, output m is v_0 + sum_k cos(2 pi m k / R) a_k -+ i sum_k sin(2 pi m k / R) b_k
// and output R - m its mirror: (R - 1)^2 / 2 real-by-complex multiply-adds instead of (R - 1)^2
// complex products, every coefficient an immediate.
template <int R, bool INV, int M, int K = 1>
__device__ __forceinline__ void dft_odd_terms(const double2 (&a)[R / 2], const double2 (&b)[R / 2], double2& cr, double2& ci) {
  if constexpr (K <= R / 2) {
    constexpr double c = odd_root_cos<R>((M * K) % R), s = odd_root_sin<R>((M * K) % R);
    cr.x = __builtin_fma(c, a[K - 1].x, cr.x);
    cr.y = __builtin_fma(c, a[K - 1].y, cr.y);
    ci.x = __builtin_fma(s, b[K - 1].x, ci.x);
    ci.y = __builtin_fma(s, b[K - 1].y, ci.y);
    dft_odd_terms<R, INV, M, K + 1>(a, b, cr, ci);
  }
}
template <int R, bool INV, int M = 1>
__device__ __forceinline__ void dft_odd_rows(double2 v0, const double2 (&a)[R / 2], const double2 (&b)[R / 2], double2 (&v)[R]) {
  if constexpr (M <= R / 2) {
    double2 cr = v0, ci = mk<double>(0.0, 0.0);
    dft_odd_terms<R, INV, M>(a, b, cr, ci);
    // forward: cr - i ci at m, cr + i ci at R - m; the inverse swaps them
    const double2 lo = mk<double>(cr.x + ci.y, cr.y - ci.x), hi = mk<double>(cr.x - ci.y, cr.y + ci.x);
    v[M] = INV ? hi : lo;
    v[R - M] = INV ? lo : hi;
    dft_odd_rows<R, INV, M + 1>(v0, a, b, v);
  }
}
template <int R, bool INV>
__device__ __forceinline__ void dft_odd(double2 (&v)[R]) {
  static_assert(R % 2 == 1 && R >= 3, "odd radices only");
  double2 a[R / 2], b[R / 2];
  const double2 v0 = v[0];
  double2 sum = v0;
#pragma unroll
  for (int k = 1; k <= R / 2; ++k) {
    a[k - 1] = cadd(v[k], v[R - k]);
    b[k - 1] = csub(v[k], v[R - k]);
    sum = cadd(sum, a[k - 1]);
  }
  v[0] = sum;
  dft_odd_rows<R, INV>(v0, a, b, v);
}

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

// ---- transforms whose data enters from, or leaves to, global memory ---------------------------
// The first forward pass can take its inputs straight from global memory and the last inverse pass can
// hand its outputs straight to global memory: the data then crosses LDS once less in each direction
// (the column pass of the registration is bound by LDS traffic, not by HBM).
//
// First pass of a forward transform (sub-block length n = N): load(f, i) returns element i of
// transform f.  Two work items per lane and trip, all their loads issued before the first butterfly.
// barrier_first: the twiddle tables were written to LDS by this workgroup and not yet synchronised;
// the barrier then comes after the first trip's loads have been issued.  Every lane runs the same
// number of trips (the barrier is inside the loop).
template <typename T, int R, class Addr, class Load>
__device__ __forceinline__ void fft_first_pass_from(cx<T>* buf, const Addr& addr, int nfft, int N, const Twiddles<T>& tw,
                                                    int tid, int nthr, bool barrier_first, Load load) {
  const int per = N / R;  // butterflies per transform; also the element stride inside one
  const int total = nfft * per;
  constexpr int kItems = 2;
  for (int base = 0; base < total; base += kItems * nthr) {
    cx<T> v[kItems][R];
    int f[kItems], j[kItems];
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
      const int w = min(base + u * nthr + tid, total - 1);  // clamp: never branch around a load
      addr.split(w, nfft, per, f[u], j[u]);
#pragma unroll
      for (int m = 0; m < R; ++m) v[u][m] = load(f[u], j[u] + m * per);
    }
    if (barrier_first && base == 0) lds_barrier();
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
      if (base + u * nthr + tid < total) {
        Dft<T, R, false>::run(v[u]);
        if (per > 1) {
          cx<T> pw[R];
          twiddle_powers<T, R>(tw.get(j[u]), pw);
#pragma unroll
          for (int d = 1; d < R; ++d) v[u][d] = cmul(v[u][d], pw[d]);
        }
#pragma unroll
        for (int m = 0; m < R; ++m) buf[addr(f[u], j[u] + m * per)] = v[u][m];
      }
    }
  }
  if (barrier_first && total <= 0) lds_barrier();
}

// Forward transform whose first pass reads global memory through load(f, i).  Returns with the
// block synchronised.  tables_unsynchronised: see fft_first_pass_from.
template <typename T, int MAXR = 16, class Addr, class Load>
__device__ __forceinline__ void fft_forward_from(cx<T>* buf, const Addr& addr, int nfft, const FftPlanDev& plan,
                                                 const Twiddles<T>& tw, int tid, int nthr, bool tables_unsynchronised,
                                                 Load load) {
  switch (plan.radix[0]) {
    case 16:
      if constexpr (MAXR >= 16) fft_first_pass_from<T, 16>(buf, addr, nfft, plan.n, tw, tid, nthr, tables_unsynchronised, load);
      break;
    case 8: fft_first_pass_from<T, 8>(buf, addr, nfft, plan.n, tw, tid, nthr, tables_unsynchronised, load); break;
    case 4: fft_first_pass_from<T, 4>(buf, addr, nfft, plan.n, tw, tid, nthr, tables_unsynchronised, load); break;
    default: fft_first_pass_from<T, 2>(buf, addr, nfft, plan.n, tw, tid, nthr, tables_unsynchronised, load); break;
  }
  lds_barrier();
  int n = plan.n / plan.radix[0];
  for (int i = 1; i < plan.npass; ++i) {
    fft_dispatch<T, false, MAXR>(plan.radix[i], buf, addr, nfft, plan.n, n, tw, tid, nthr);
    n /= plan.radix[i];
    lds_barrier();
  }
}

// Last pass of an inverse transform (sub-block length n = N): store(f, i, value) takes element i of
// transform f.
template <typename T, int R, class Addr, class Store>
__device__ __forceinline__ void fft_last_pass_to(const cx<T>* buf, const Addr& addr, int nfft, int N, const Twiddles<T>& tw,
                                                 int tid, int nthr, Store store) {
  const int per = N / R;
  for (int w = tid; w < nfft * per; w += nthr) {
    int f, j;
    addr.split(w, nfft, per, f, j);
    cx<T> v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = buf[addr(f, j + m * per)];
    if (per > 1) {
      cx<T> pw[R];
      twiddle_powers<T, R>(tw.get(j), pw);
#pragma unroll
      for (int d = 1; d < R; ++d) v[d] = cmulc(v[d], pw[d]);
    }
    Dft<T, R, true>::run(v);
#pragma unroll
    for (int m = 0; m < R; ++m) store(f, j + m * per, v[m]);
  }
}

// Unnormalised inverse transform whose last pass hands its results to store(f, i, value) instead of
// writing them back to LDS.  No barrier after the last pass: it writes no LDS.
template <typename T, int MAXR = 16, class Addr, class Store>
__device__ __forceinline__ void fft_inverse_to(cx<T>* buf, const Addr& addr, int nfft, const FftPlanDev& plan,
                                               const Twiddles<T>& tw, int tid, int nthr, Store store) {
  int n = 1;
  for (int i = plan.npass - 1; i >= 1; --i) {
    n *= plan.radix[i];
    fft_dispatch<T, true, MAXR>(plan.radix[i], buf, addr, nfft, plan.n, n, tw, tid, nthr);
    lds_barrier();
  }
  switch (plan.radix[0]) {
    case 16:
      if constexpr (MAXR >= 16) fft_last_pass_to<T, 16>(buf, addr, nfft, plan.n, tw, tid, nthr, store);
      break;
    case 8: fft_last_pass_to<T, 8>(buf, addr, nfft, plan.n, tw, tid, nthr, store); break;
    case 4: fft_last_pass_to<T, 4>(buf, addr, nfft, plan.n, tw, tid, nthr, store); break;
    default: fft_last_pass_to<T, 2>(buf, addr, nfft, plan.n, tw, tid, nthr, store); break;
  }
}

// ---- lengths with odd factors ------------------------------------------------------------------
// The same in-place scheme for a length N = product of radices from {2, 3, 4, 5, 7, 8, 11, 13}
// (complex128, transforms as columns of a row-major [N][rowstride] matrix, a power-of-two number
// of them).  Block lengths are no longer powers of two, so butterflies are located by a division
// (a multiply-high by a precomputed reciprocal) and twiddles come from the whole table of N roots in
// LDS: the index j d N / n of the d-th twiddle of butterfly j never reaches N, so no reduction either.
template <int R, bool INV>
__device__ __forceinline__ void mixed_pass(double2* buf, int rowstride, int lncol, int N, int n, const double2* twN,
                                           int tid, int nthr) {
  const int per = N / R, sub = n / R, twstep = N / n;
  const unsigned magic = sub > 1 ? 0xffffffffu / (unsigned)sub + 1u : 0u;  // exact quotients while g * sub < 2^32
  const int cmask = (1 << lncol) - 1;
  for (int w = tid; w < (per << lncol); w += nthr) {
    const int g = w >> lncol, f = w & cmask;
    const int blk = sub > 1 ? (int)__umulhi((unsigned)g, magic) : g;
    const int j = g - blk * sub;
    double2* at = buf + (blk * n + j) * rowstride + f;
    const int step = sub * rowstride;
    double2 v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = at[m * step];
    if constexpr (!INV) {
      if constexpr (R & 1)
        dft_odd<R, false>(v);
      else
        Dft<double, R, false>::run(v);
      if (sub > 1) {
        const int t = j * twstep;
#pragma unroll
        for (int d = 1; d < R; ++d) v[d] = cmul(v[d], twN[t * d]);
      }
    } else {
      if (sub > 1) {
        const int t = j * twstep;
#pragma unroll
        for (int d = 1; d < R; ++d) v[d] = cmulc(v[d], twN[t * d]);
      }
      if constexpr (R & 1)
        dft_odd<R, true>(v);
      else
        Dft<double, R, true>::run(v);
    }
#pragma unroll
    for (int m = 0; m < R; ++m) at[m * step] = v[m];
  }
}

// The same pass with its inputs taken through in(f, i) and its results handed to out(f, i, value)
// (f the column, i the element): the first forward pass reads global memory and the last inverse pass
// writes it, as fft_first_pass_from / fft_last_pass_to do for the power-of-two lengths.  Every lane runs the
// same number of trips; with barrier_first the workgroup synchronises once, after the first trip's inputs have
// been requested (the table of roots was written to LDS by this workgroup just before).
template <int R, bool INV, class In, class Out>
__device__ __forceinline__ void mixed_pass_io(int lncol, int N, int n, const double2* twN, int tid, int nthr,
                                              bool barrier_first, In in, Out out) {
  const int per = N / R, sub = n / R, twstep = N / n;
  const unsigned magic = sub > 1 ? 0xffffffffu / (unsigned)sub + 1u : 0u;
  const int cmask = (1 << lncol) - 1, total = per << lncol;
  for (int w0 = 0; w0 < total; w0 += nthr) {
    const int w = min(w0 + tid, total - 1);  // clamp: never branch around a load
    const int g = w >> lncol, f = w & cmask;
    const int blk = sub > 1 ? (int)__umulhi((unsigned)g, magic) : g;
    const int j = g - blk * sub;
    const int base = blk * n + j;
    double2 v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = in(f, base + m * sub);
    if (barrier_first && w0 == 0) lds_barrier();
    if (w0 + tid < total) {
      if constexpr (!INV) {
        if constexpr (R & 1)
          dft_odd<R, false>(v);
        else
          Dft<double, R, false>::run(v);
        if (sub > 1) {
          const int t = j * twstep;
#pragma unroll
          for (int d = 1; d < R; ++d) v[d] = cmul(v[d], twN[t * d]);
        }
      } else {
        if (sub > 1) {
          const int t = j * twstep;
#pragma unroll
          for (int d = 1; d < R; ++d) v[d] = cmulc(v[d], twN[t * d]);
        }
        if constexpr (R & 1)
          dft_odd<R, true>(v);
        else
          Dft<double, R, true>::run(v);
      }
#pragma unroll
      for (int m = 0; m < R; ++m) out(f, base + m * sub, v[m]);
    }
  }
}

template <bool INV, class In, class Out>
__device__ __forceinline__ void mixed_dispatch_io(int radix, int lncol, int N, int n, const double2* twN, int tid, int nthr,
                                                  bool barrier_first, In in, Out out) {
  switch (radix) {
    case 13: mixed_pass_io<13, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    case 11: mixed_pass_io<11, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    case 8: mixed_pass_io<8, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    case 7: mixed_pass_io<7, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    case 5: mixed_pass_io<5, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    case 4: mixed_pass_io<4, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    case 3: mixed_pass_io<3, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
    default: mixed_pass_io<2, INV>(lncol, N, n, twN, tid, nthr, barrier_first, in, out); break;
  }
}

template <bool INV>
__device__ __forceinline__ void mixed_dispatch(int radix, double2* buf, int rowstride, int lncol, int N, int n,
                                               const double2* twN, int tid, int nthr) {
  switch (radix) {
    case 13: mixed_pass<13, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    case 11: mixed_pass<11, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    case 8: mixed_pass<8, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    case 7: mixed_pass<7, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    case 5: mixed_pass<5, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    case 4: mixed_pass<4, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    case 3: mixed_pass<3, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
    default: mixed_pass<2, INV>(buf, rowstride, lncol, N, n, twN, tid, nthr); break;
  }
}

// Forward: natural order in, digit-reversed out.  Returns with the block synchronised.
__device__ __forceinline__ void mixed_forward(double2* buf, int rowstride, int lncol, const FftPlanDev& plan,
                                              const double2* twN, int tid, int nthr) {
  int n = plan.n;
  for (int i = 0; i < plan.npass; ++i) {
    mixed_dispatch<false>(plan.radix[i], buf, rowstride, lncol, plan.n, n, twN, tid, nthr);
    n /= plan.radix[i];
    lds_barrier();
  }
}

// Unnormalised inverse: digit-reversed in, natural out.
__device__ __forceinline__ void mixed_inverse(double2* buf, int rowstride, int lncol, const FftPlanDev& plan,
                                              const double2* twN, int tid, int nthr) {
  int n = 1;
  for (int i = plan.npass - 1; i >= 0; --i) {
    n *= plan.radix[i];
    mixed_dispatch<true>(plan.radix[i], buf, rowstride, lncol, plan.n, n, twN, tid, nthr);
    lds_barrier();
  }
}

// Forward transform whose first pass takes its inputs through load(f, i) (global memory); tables_unsynchronised:
// the roots were written to LDS by this workgroup and no barrier has followed yet.
template <class Load>
__device__ __forceinline__ void mixed_forward_from(double2* buf, int rowstride, int lncol, const FftPlanDev& plan,
                                                   const double2* twN, int tid, int nthr, bool tables_unsynchronised,
                                                   Load load) {
  mixed_dispatch_io<false>(plan.radix[0], lncol, plan.n, plan.n, twN, tid, nthr, tables_unsynchronised, load,
                           [&](int f, int i, double2 v) { buf[i * rowstride + f] = v; });
  lds_barrier();
  int n = plan.n / plan.radix[0];
  for (int i = 1; i < plan.npass; ++i) {
    mixed_dispatch<false>(plan.radix[i], buf, rowstride, lncol, plan.n, n, twN, tid, nthr);
    n /= plan.radix[i];
    lds_barrier();
  }
}

// Unnormalised inverse whose last pass hands its results to store(f, i, value) and writes no LDS.
template <class Store>
__device__ __forceinline__ void mixed_inverse_to(double2* buf, int rowstride, int lncol, const FftPlanDev& plan,
                                                 const double2* twN, int tid, int nthr, Store store) {
  int n = 1;
  for (int i = plan.npass - 1; i >= 1; --i) {
    n *= plan.radix[i];
    mixed_dispatch<true>(plan.radix[i], buf, rowstride, lncol, plan.n, n, twN, tid, nthr);
    lds_barrier();
  }
  mixed_dispatch_io<true>(plan.radix[0], lncol, plan.n, plan.n, twN, tid, nthr, false,
                          [&](int f, int i) { return buf[i * rowstride + f]; }, store);
}


}  // namespace mistitch
