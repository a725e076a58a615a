// Register-resident complex128 transforms of power-of-two lengths N = 512 R0, R0 in {1, 2, 4, 8}.
//
// The in-LDS engine of fft.hpp reads and writes the whole line once per radix pass, with run-time strides.
// Counters on the 2048-point lines showed its cost to be the SUM of its vector arithmetic (half of it address
// arithmetic) and its LDS traffic (a third of it bank conflicts), not their maximum.  Here a lane keeps its 8
// elements in registers through every radix pass, every LDS offset is a compile-time immediate, and LDS only
// carries the exchanges between passes: one across the workgroup (after the first pass, radix R0, which leaves
// R0 independent 512-point blocks) and, inside each wave, which owns one such block, the exchanges of its three
// radix-8 passes, which need no workgroup barrier.  Same decimation-in-frequency scheme, same twiddles and the
// same digit-reversed storage positions as fft_pass with the plan (R0, 8, 8, 8): every table of the plan stays
// valid.
//
// LDS image of a line: element p at slot p + (p >> 3) (one slot of padding per 8 elements), which makes the
// stride-8 and stride-64 accesses of the late passes conflict free for 16-byte reads and writes.
#pragma once

#include "fft.hpp"

namespace mistitch {

template <int N>
struct RegFft {
  static_assert(N == 512 || N == 1024 || N == 2048 || N == 4096, "512 R0 points, R0 in {1, 2, 4, 8}");
  static constexpr int NL = N / 8;       // lanes per line
  static constexpr int R0 = N / 512;     // radix of the first (forward) pass; waves per line
  static constexpr int NB = 8 / R0;      // first-pass butterflies per lane
  static constexpr int kSlots = N + N / 8;  // LDS slots of a line
  __device__ __forceinline__ static int slot(int p) { return p + (p >> 3); }

  // Twiddle bases of one lane: first pass (butterflies t + k NL), second (index l of the wave's 512-point block),
  // third (index l & 7 of its 64-point block); tw = all N roots exp(-2 pi i k / N).
  struct Tw {
    double2 p0[NB];
    double2 p1, p2;
  };
  __device__ __forceinline__ static Tw load_twiddles(const double2* __restrict__ tw, int t) {
    Tw r;
    const int l = t & 63;
#pragma unroll
    for (int k = 0; k < NB; ++k) r.p0[k] = tw[R0 > 1 ? t + k * NL : 0];
    r.p1 = tw[l * R0];
    r.p2 = tw[(l & 7) * (N / 64)];
    return r;
  }

  template <int R, bool INV>
  __device__ __forceinline__ static void butterfly(double2 (&x)[R], double2 base, bool twiddled) {
    if constexpr (!INV) {
      Dft<double, R, false>::run(x);
      if (twiddled) {
        double2 pw[R];
        twiddle_powers<double, R>(base, pw);
#pragma unroll
        for (int d = 1; d < R; ++d) x[d] = cmul(x[d], pw[d]);
      }
    } else {
      if (twiddled) {
        double2 pw[R];
        twiddle_powers<double, R>(base, pw);
#pragma unroll
        for (int d = 1; d < R; ++d) x[d] = cmulc(x[d], pw[d]);
      }
      Dft<double, R, true>::run(x);
    }
  }

  // Compiler-level ordering of a wave's own LDS writes before its following reads (the hardware executes one
  // wave's LDS instructions in order).
  __device__ __forceinline__ static void wave_order() { asm volatile("" ::: "memory"); }

  // Forward transform.  On entry lane t holds elements t + NL j (natural order) in v[j]; on return the wave's
  // block w = t / 64 lies in LDS in storage order, element w 512 + q at slot(w 512 + q), written by this wave only
  // (read it back after wave_order(), no barrier).  One workgroup barrier inside (none for N = 512).
  __device__ __forceinline__ static void forward(double2 (&v)[8], double2* buf, const Tw& tw, int t) {
    const int w = t >> 6, l = t & 63;
    if constexpr (R0 > 1) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        double2 x[R0];
#pragma unroll
        for (int m = 0; m < R0; ++m) x[m] = v[k + NB * m];
        butterfly<R0, false>(x, tw.p0[k], true);
#pragma unroll
        for (int m = 0; m < R0; ++m) v[k + NB * m] = x[m];
      }
      double2* at = buf + slot(t);
#pragma unroll
      for (int j = 0; j < 8; ++j) at[j * (NL + NL / 8)] = v[j];
      lds_barrier();
      const double2* from = buf + slot(w * 512 + l);
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = from[72 * m];
    }
    butterfly<8, false>(v, tw.p1, true);
    {
      double2* at = buf + slot(w * 512 + l);
#pragma unroll
      for (int m = 0; m < 8; ++m) at[72 * m] = v[m];
      wave_order();
      const double2* from = buf + slot(w * 512 + (l >> 3) * 64 + (l & 7));
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = from[9 * m];
    }
    butterfly<8, false>(v, tw.p2, true);
    {
      double2* at = buf + slot(w * 512 + (l >> 3) * 64 + (l & 7));
#pragma unroll
      for (int m = 0; m < 8; ++m) at[9 * m] = v[m];
      wave_order();
      const double2* from = buf + w * 576 + 9 * l;
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = from[m];
    }
    butterfly<8, false>(v, mk<double>(1.0, 0.0), false);
    {
      double2* at = buf + w * 576 + 9 * l;
#pragma unroll
      for (int m = 0; m < 8; ++m) at[m] = v[m];
      wave_order();
    }
  }

  // Unnormalised inverse transform (the adjoint passes in reverse).  On entry the line lies in LDS in storage
  // order (element p at slot(p)) and the workgroup is synchronised; on return lane t holds the natural-order
  // elements t + NL j in v[j].  One workgroup barrier inside (none for N = 512).
  __device__ __forceinline__ static void inverse(double2 (&v)[8], double2* buf, const Tw& tw, int t) {
    const int w = t >> 6, l = t & 63;
    {
      const double2* from = buf + w * 576 + 9 * l;
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = from[m];
    }
    butterfly<8, true>(v, mk<double>(1.0, 0.0), false);
    {
      double2* at = buf + w * 576 + 9 * l;
#pragma unroll
      for (int m = 0; m < 8; ++m) at[m] = v[m];
      wave_order();
      const double2* from = buf + slot(w * 512 + (l >> 3) * 64 + (l & 7));
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = from[9 * m];
    }
    butterfly<8, true>(v, tw.p2, true);
    {
      double2* at = buf + slot(w * 512 + (l >> 3) * 64 + (l & 7));
#pragma unroll
      for (int m = 0; m < 8; ++m) at[9 * m] = v[m];
      wave_order();
      const double2* from = buf + slot(w * 512 + l);
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = from[72 * m];
    }
    butterfly<8, true>(v, tw.p1, true);
    if constexpr (R0 > 1) {
      double2* at = buf + slot(w * 512 + l);
#pragma unroll
      for (int m = 0; m < 8; ++m) at[72 * m] = v[m];
      lds_barrier();
      const double2* from = buf + slot(t);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = from[j * (NL + NL / 8)];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        double2 x[R0];
#pragma unroll
        for (int m = 0; m < R0; ++m) x[m] = v[k + NB * m];
        butterfly<R0, true>(x, tw.p0[k], true);
#pragma unroll
        for (int m = 0; m < R0; ++m) v[k + NB * m] = x[m];
      }
    }
  }
};

}  // namespace mistitch
