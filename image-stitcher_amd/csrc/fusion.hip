// Tile fusion, flatfield application and z-projection kernels with their C
// entry points.  Everything here is a streaming pass: one read of each
// winning source pixel, one write of each mosaic pixel, 16 bytes per lane.
//
// Fusion is a gather over the OUTPUT: a workgroup owns a kBH x kBW pixel block
// of one mosaic plane, collects (in placement order) the tiles that touch the
// block into LDS, and every lane resolves one 8-pixel window per row.  Writing
// each output pixel exactly once makes "the last placed tile wins" a per-pixel
// rule that is independent of how the mosaic is cut into blocks, bands or GPUs.
#include <algorithm>

#include "common.hpp"

namespace mistitch {

constexpr int kFuseThreads = 256;
#ifndef MS_FUSE_GX_PLAIN
#define MS_FUSE_GX_PLAIN 4
#endif
#ifndef MS_FUSE_GX_FF
#define MS_FUSE_GX_FF 4
#endif
#ifndef MS_FUSE_BH
#define MS_FUSE_BH 32
#endif
constexpr int kGxPlain = MS_FUSE_GX_PLAIN;  // groups per lane and row without flatfield (block width = 512 * GX px)
constexpr int kGxFlat = MS_FUSE_GX_FF;      // with flatfield: fewer, the flatfield values take registers too
constexpr int kBH = MS_FUSE_BH;             // block height in pixels
constexpr int kMaxCand = 32;  // tiles remembered per block before the slow path
// Waves of a workgroup side by side along a row (1: each of the four waves takes its own rows of a 512 GX pixel wide
// block; 4: the four waves take four neighbouring 512 GX pixel pieces of the SAME row, so that the flatfield row they
// all read - neighbouring tiles of a grid row use the same one - is fetched into the CU's L1 once).  The queue entries
// hold 13 bits of (group in the block row, row in the block): four waves across need blocks of 8 rows.
#ifndef MS_FUSE_WX
#define MS_FUSE_WX 1
#endif
constexpr int kWX = MS_FUSE_WX, kWY = (kFuseThreads / 64) / kWX;
static_assert(kWX == 1 || kWX == 2 || kWX == 4, "waves across");
static_assert(kBH * kWX <= 32 && kMaxCand == 32, "queue entries hold 13 bits of group and row, and 5-bit candidate indices");

#ifndef MS_FDIV
#define MS_FDIV(a, b) __fdiv_rn((a), (b))
#endif

struct __attribute__((packed, aligned(2))) Px8 {
  unsigned d[4];  // 8 pixels, two per dword
  __device__ __forceinline__ unsigned at(int i) const { return (d[i >> 1] >> (16 * (i & 1))) & 0xffffu; }
  __device__ __forceinline__ void set(int i, unsigned px) {
    d[i >> 1] = (i & 1) ? ((d[i >> 1] & 0xffffu) | (px << 16)) : ((d[i >> 1] & 0xffff0000u) | px);
  }
};
#define MS_PIN_PX8(p) \
  do {                \
    MS_PIN((p).d[0]); \
    MS_PIN((p).d[1]); \
    MS_PIN((p).d[2]); \
    MS_PIN((p).d[3]); \
  } while (0)
#define MS_PIN_8(p)                       \
  do {                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) MS_PIN((p).v[i_]); \
  } while (0)
// Mosaic pixels are written once and never read back: stored non-temporal they leave the caches to
// the flatfield planes, which every tile reads again (2 % with a flatfield, nothing without).  The
// same hint on the tile pixel loads costs time: neighbouring lanes' unaligned 16-byte reads share lines.
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_px8_stream(uint16_t* p, const Px8& v) {  // p 16-byte aligned
  __builtin_nontemporal_store(nt_u4{v.d[0], v.d[1], v.d[2], v.d[3]}, reinterpret_cast<nt_u4*>(p));
}

struct __attribute__((packed, aligned(4))) F8 {
  float v[8];
};
struct __attribute__((packed, aligned(8))) D8 {
  double v[8];
};

struct TileRef {
  int ty, tx;        // mosaic position of the tile origin
  int t;             // tile index
  int ff;            // flatfield plane or -1
  int64_t src_base;  // element offset, in the tile array, of the tile pixel under mosaic pixel (y, x) is src_base + y * W + x
  int64_t ff_base;   // the same in the flatfield array (0 without a plane)
};

__device__ __forceinline__ TileRef make_ref(int t, const int32_t* __restrict__ tile_yx, const int32_t* tile_ff, int H,
                                            int W, int64_t tile_pitch) {
  TileRef c;
  c.ty = tile_yx[2 * t];
  c.tx = tile_yx[2 * t + 1];
  c.t = t;
  c.ff = tile_ff ? tile_ff[t] : -1;
  const int64_t origin = (int64_t)c.ty * W + c.tx;
  c.src_base = (int64_t)t * tile_pitch - origin;
  c.ff_base = c.ff >= 0 ? (int64_t)c.ff * H * W - origin : 0;
  return c;
}

template <int FF>
__device__ __forceinline__ uint16_t flatfield_one(uint16_t v, const void* ffrow, int i, unsigned vmax) {
  if constexpr (FF == MISTITCH_FF_F32) {
    float q = MS_FDIV((float)v, reinterpret_cast<const float*>(ffrow)[i]);
    q = fminf(fmaxf(q, 0.0f), (float)vmax);
    return (uint16_t)(unsigned int)q;
  } else if constexpr (FF == MISTITCH_FF_F64) {
    double q = (double)v / reinterpret_cast<const double*>(ffrow)[i];
    q = fmin(fmax(q, 0.0), (double)vmax);
    return (uint16_t)(unsigned int)q;
  } else {
    return v;
  }
}

// Fetch the pixels of `m` (bit i = window pixel i) from tile row `src` (points
// at window pixel 0, which may lie outside the tile when bit 0 is clear).
template <int FF>
__device__ __forceinline__ void fetch_window(const uint16_t* src, const void* ffrow, unsigned m, uint16_t (&px)[8],
                                             unsigned vmax) {
  if (m == 0xffu) {
    const Px8 v = *reinterpret_cast<const Px8*>(src);
    if constexpr (FF == MISTITCH_FF_F32) {
      const F8 f = *reinterpret_cast<const F8*>(ffrow);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float q = MS_FDIV((float)v.at(i), f.v[i]);
        q = fminf(fmaxf(q, 0.0f), (float)vmax);
        px[i] = (uint16_t)(unsigned int)q;
      }
    } else if constexpr (FF == MISTITCH_FF_F64) {
      const D8 f = *reinterpret_cast<const D8*>(ffrow);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double q = (double)v.at(i) / f.v[i];
        q = fmin(fmax(q, 0.0), (double)vmax);
        px[i] = (uint16_t)(unsigned int)q;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) px[i] = (uint16_t)v.at(i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (m & (1u << i)) px[i] = flatfield_one<FF>(src[i], ffrow, i, vmax);
  }
}

// Output blocks are cut in ADDRESS space, not pixel space: lane l of block bx owns,
// in every row, the 16-byte-aligned group of 8 mosaic elements number bx * 64 + l
// (and the next kGX - 1 groups at a stride of 64) counted from the aligned
// address at or below the row start.  Mosaic rows have
// arbitrary (odd) lengths, so the pixel column of a group shifts by up to 7 from
// row to row; in exchange every full group is one aligned 16-byte store and a wave
// writes whole cache lines, kGX KiB of one row at a time.  A block's pixel columns lie in
// [bx * BW - 7, bx * BW + BW) with BW = 512 * GX.

// One wave lists, in placement order, the tiles that touch each of kBinRows vertically adjacent output blocks
// (every tile's position is loaded once per wave, not once per block):
// bins[block][0] = how many (may exceed kMaxCand: the fuse kernel then scans every
// tile), bins[block][1..] = the first kMaxCand tile indices.
// grid = ceil(n_groups / 4) workgroups of 4 waves, n_groups = gx * ceil(gy / kBinRows) * n_planes; blocks are
// numbered as k_fuse's grid.
constexpr int kBinRows = 8;
__global__ __launch_bounds__(256) void k_fuse_bins(int n_tiles, int H, int W, const int32_t* __restrict__ tile_yx,
                                                   const int32_t* __restrict__ tile_plane, int32_t* __restrict__ bins,
                                                   int gx, int gy, int n_planes, int Wm, int row0, int row1, int kBW,
                                                   const int32_t* __restrict__ tile_ff, int32_t* ff_verdict) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0) {
    // For k_ff_check, next on the stream: ff_verdict[0] = 1 (it may clear it; a flatfield plane index beyond
    // kFfPlanes clears it here), ff_verdict[1..2] = which of the planes 0..63 some tile uses, one bit each.
    __shared__ unsigned part[4][3];
    unsigned lo = 0, hi = 0, beyond = 0;
    for (int t = threadIdx.x; tile_ff && t < n_tiles; t += blockDim.x) {
      const int f = tile_ff[t];  // negative: the tile has no plane
      lo |= f >= 0 && f < 32 ? 1u << f : 0u;
      hi |= f >= 32 && f < 64 ? 1u << (f - 32) : 0u;
      beyond |= f >= 64;
    }
    for (int s = 1; s < 64; s <<= 1) {
      lo |= __shfl_xor(lo, s);
      hi |= __shfl_xor(hi, s);
      beyond |= __shfl_xor(beyond, s);
    }
    if (lane == 0) {
      part[threadIdx.x >> 6][0] = lo;
      part[threadIdx.x >> 6][1] = hi;
      part[threadIdx.x >> 6][2] = beyond;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      ff_verdict[0] = (part[0][2] | part[1][2] | part[2][2] | part[3][2]) ? 0 : 1;
      ff_verdict[1] = (int32_t)(part[0][0] | part[1][0] | part[2][0] | part[3][0]);
      ff_verdict[2] = (int32_t)(part[0][1] | part[1][1] | part[2][1] | part[3][1]);
    }
  }
  const int gyg = (gy + kBinRows - 1) / kBinRows;
  const int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grp >= (int64_t)gx * gyg * n_planes) return;
  const int bx = (int)(grp % gx), byg = (int)((grp / gx) % gyg), plane = (int)(grp / ((int64_t)gx * gyg));
  const int x0 = max(bx * kBW - 7, 0), x1 = min(bx * kBW + kBW, Wm);
  const int by0 = byg * kBinRows, nby = min(kBinRows, gy - by0);
  const int ya = row0 + by0 * kBH, yb = min(ya + nby * kBH, row1);  // rows of the whole group
  int count[kBinRows];
#pragma unroll
  for (int k = 0; k < kBinRows; ++k) count[k] = 0;
  for (int base = 0; base < n_tiles; base += 64) {
    const int t = base + lane;
    int ty = 0, tx = 0;
    bool near = false;
    if (t < n_tiles && tile_plane[t] == plane) {
      ty = tile_yx[2 * t];
      tx = tile_yx[2 * t + 1];
      near = ty < yb && ty + H > ya && tx < x1 && tx + W > x0;
    }
    if (__ballot(near) == 0ull) continue;  // wave-uniform
#pragma unroll
    for (int k = 0; k < kBinRows; ++k) {
      if (k < nby) {
        const int y0 = ya + k * kBH, y1 = min(y0 + kBH, row1);
        const bool hit = near && ty < y1 && ty + H > y0;
        const unsigned long long mask = __ballot(hit);
        if (hit) {
          const int slot = count[k] + __popcll(mask & ((1ull << lane) - 1ull));
          if (slot < kMaxCand) bins[(((int64_t)plane * gy + by0 + k) * gx + bx) * (kMaxCand + 1) + 1 + slot] = t;
        }
        count[k] += __popcll(mask);
      }
    }
  }
  if (lane < nby) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < kBinRows; ++k) c = lane == k ? count[k] : c;
    bins[(((int64_t)plane * gy + by0 + lane) * gx + bx) * (kMaxCand + 1)] = c;
  }
}

__device__ __forceinline__ void store_window(uint16_t* dst, const uint16_t (&px)[8], unsigned valid) {
  if (valid == 0xffu) {
    uint4 o;
    o.x = px[0] | ((unsigned)px[1] << 16);
    o.y = px[2] | ((unsigned)px[3] << 16);
    o.z = px[4] | ((unsigned)px[5] << 16);
    o.w = px[6] | ((unsigned)px[7] << 16);
    *reinterpret_cast<uint4*>(dst) = o;  // dst is 16-byte aligned by construction
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (valid & (1u << i)) dst[i] = px[i];
  }
}

// Resolve one 8-pixel window at (y, x..x+7) against the candidate list.
template <int MODE, int FF>
__device__ __forceinline__ void resolve_window(int y, int x, unsigned valid, const TileRef* list, int ncand,
                                               bool overflow, int plane, const uint16_t* __restrict__ tiles, int H,
                                               int W, int64_t tile_pitch, const int32_t* __restrict__ tile_yx,
                                               const int32_t* __restrict__ tile_plane,
                                               const void* __restrict__ flatfields,
                                               const int32_t* __restrict__ tile_ff, uint16_t (&px)[8], unsigned vmax) {
  constexpr size_t ffsize = FF == MISTITCH_FF_F64 ? 8 : 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) px[i] = 0;
  if constexpr (MODE == MISTITCH_FUSE_OVERWRITE) {
    unsigned todo = valid;
    for (int ci = ncand - 1; ci >= 0 && todo; --ci) {
      TileRef c;
      if (!overflow) {
        c = list[ci];
      } else {
        if (tile_plane[ci] != plane) continue;
        c = make_ref(ci, tile_yx, tile_ff, H, W, tile_pitch);
      }
      const int ly = y - c.ty;
      if ((unsigned)ly >= (unsigned)H) continue;
      const int lx0 = x - c.tx;
      const int lo = max(0, -lx0), hi = min(8, W - lx0);
      if (lo >= hi) continue;
      const unsigned m = (((1u << hi) - 1u) & ~((1u << lo) - 1u)) & todo;
      if (!m) continue;
      const uint16_t* src = tiles + (int64_t)c.t * tile_pitch + (int64_t)ly * W + lx0;
      if (FF != MISTITCH_FF_NONE && c.ff >= 0) {
        const char* ffrow =
            reinterpret_cast<const char*>(flatfields) + (((int64_t)c.ff * H + ly) * W + lx0) * (int64_t)ffsize;
        fetch_window<FF>(src, ffrow, m, px, vmax);
      } else {
        fetch_window<MISTITCH_FF_NONE>(src, nullptr, m, px, vmax);
      }
      todo &= ~m;
    }
  } else {
    // Feather: weighted mean with the separable ramp w = min(ly+1, H-ly) * min(lx+1, W-lx),
    // float32 operations rounded one by one in placement order.
    float num[8], den[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) num[i] = den[i] = 0.0f;
    int n_touch = 0;
    for (int ci = 0; ci < ncand; ++ci) {
      TileRef c;
      if (!overflow) {
        c = list[ci];
      } else {
        if (tile_plane[ci] != plane) continue;
        c = make_ref(ci, tile_yx, tile_ff, H, W, tile_pitch);
      }
      const int ly = y - c.ty;
      if ((unsigned)ly >= (unsigned)H) continue;
      const int lx0 = x - c.tx;
      const int lo = max(0, -lx0), hi = min(8, W - lx0);
      if (lo >= hi) continue;
      const unsigned m = (((1u << hi) - 1u) & ~((1u << lo) - 1u)) & valid;
      if (!m) continue;
      uint16_t v[8];
      const uint16_t* src = tiles + (int64_t)c.t * tile_pitch + (int64_t)ly * W + lx0;
      if (FF != MISTITCH_FF_NONE && c.ff >= 0) {
        const char* ffrow =
            reinterpret_cast<const char*>(flatfields) + (((int64_t)c.ff * H + ly) * W + lx0) * (int64_t)ffsize;
        fetch_window<FF>(src, ffrow, m, v, vmax);
      } else {
        fetch_window<MISTITCH_FF_NONE>(src, nullptr, m, v, vmax);
      }
      const float wy = (float)min(ly + 1, H - ly);
      ++n_touch;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (m & (1u << i)) {
          const int lx = lx0 + i;
          const float w = __fmul_rn(wy, (float)min(lx + 1, W - lx));
          num[i] = __fadd_rn(num[i], __fmul_rn(w, (float)v[i]));
          den[i] = __fadd_rn(den[i], w);
          if (n_touch == 1) px[i] = v[i];
        }
      }
    }
    if (n_touch > 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (den[i] > 0.0f) {
          const float q = floorf(__fadd_rn(MS_FDIV(num[i], den[i]), 0.5f));
          px[i] = (uint16_t)(unsigned int)fminf(q, 65535.0f);
        }
      }
    }
  }
}

// ---- two pixels per instruction -------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// n / d, correctly rounded, for operands that need no range scaling (d a normal
// number of moderate exponent, the quotient neither overflowing nor denormal).
// It is the fma chain of the IEEE-exact division expansion (v_div_scale ...
// v_div_fixup) with the scaling steps, which are the identity for such operands,
// left out, on two quotients at once (v_pk_fma_f32).
__device__ __forceinline__ f2 div2_noscale(f2 n, f2 d) {
  f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 one = {1.0f, 1.0f};
  const f2 e = pk_fma(-d, r, one);
  r = pk_fma(e, r, r);
  f2 q = n * r;
  f2 t = pk_fma(-d, q, n);
  q = pk_fma(t, r, q);
  t = pk_fma(-d, q, n);
  return pk_fma(t, r, q);
}

// n / d for an integer numerator 0..65535 and a denominator within [2^-16, 2^16]:
// one Newton step on the hardware reciprocal and one residual correction give the
// correctly rounded quotient for every such pair (checked exhaustively on gfx950 over
// all 2^16 numerators x 2^23 denominator mantissas by tools/div_check.hip; powers of
// two scale every intermediate exactly).
__device__ __forceinline__ f2 div2_u16(f2 n, f2 d) {
  f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 one = {1.0f, 1.0f};
  const f2 e = pk_fma(-d, r, one);
  r = pk_fma(e, r, r);
  const f2 q = n * r;
  const f2 t = pk_fma(-d, q, n);
  return pk_fma(t, r, q);
}

// Flatfield-corrected values of 8 pixels, trunc(clip(v / ff)), as unsigned integers in
// [0, 65535].  SAFE: every flatfield value is known to be finite and within
// [2^-16, 2^16] (checked once per call by k_ff_check), which admits div2_u16.
template <int FF, bool SAFE>
__device__ __forceinline__ void corrected8(const Px8& v, const F8& f32, const D8& f64, unsigned (&out)[8], unsigned vmax) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned d = v.d[p];
    if constexpr (FF == MISTITCH_FF_F32) {
      const f2 n = {(float)(d & 0xffffu), (float)(d >> 16)};
      if constexpr (SAFE) {
        const f2 q = div2_u16(n, f2{f32.v[2 * p], f32.v[2 * p + 1]});  // positive; saturated when packed or below
        out[2 * p] = min((unsigned int)q.x, vmax);
        out[2 * p + 1] = min((unsigned int)q.y, vmax);
      } else {
        out[2 * p] = (unsigned int)fminf(fmaxf(MS_FDIV(n.x, f32.v[2 * p]), 0.0f), (float)vmax);
        out[2 * p + 1] = (unsigned int)fminf(fmaxf(MS_FDIV(n.y, f32.v[2 * p + 1]), 0.0f), (float)vmax);
      }
    } else if constexpr (FF == MISTITCH_FF_F64) {
      out[2 * p] = (unsigned int)fmin(fmax((double)(d & 0xffffu) / f64.v[2 * p], 0.0), (double)vmax);
      out[2 * p + 1] = (unsigned int)fmin(fmax((double)(d >> 16) / f64.v[2 * p + 1], 0.0), (double)vmax);
    } else {
      out[2 * p] = d & 0xffffu;
      out[2 * p + 1] = d >> 16;
    }
  }
}

// Two values to one dword of pixels; values above 65535 saturate.
__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const us2 pk = __builtin_amdgcn_cvt_pk_u16(lo, hi);
  return __builtin_bit_cast(unsigned, pk);
}

template <int FF, bool SAFE>
__device__ __forceinline__ Px8 corrected8_packed(const Px8& v, const F8& f32, const D8& f64, unsigned vmax) {
  if constexpr (FF == MISTITCH_FF_F32 && SAFE) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 top = {(unsigned short)vmax, (unsigned short)vmax};
    Px8 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned d = v.d[p];
      const f2 n = {(float)(d & 0xffffu), (float)(d >> 16)};
#if defined(MS_FUSE_ABLATE) && MS_FUSE_ABLATE == 1   /* measurement build: a product in place of the quotient */
      const f2 q = n * f2{f32.v[2 * p], f32.v[2 * p + 1]};
#elif defined(MS_FUSE_ABLATE) && MS_FUSE_ABLATE == 2  /* measurement build: no arithmetic on the pixels at all */
      const f2 q = {__uint_as_float(d) + f32.v[2 * p], f32.v[2 * p + 1]};
#else
      const f2 q = div2_u16(n, f2{f32.v[2 * p], f32.v[2 * p + 1]});
#endif
      o.d[p] = pack2((unsigned int)q.x, (unsigned int)q.y);  // saturates at 65535
    }
    if (vmax != 65535u) {  // (the same for every lane) 8-bit tiles: the pixel type's largest value is smaller
#pragma unroll
      for (int p = 0; p < 4; ++p)
        o.d[p] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, o.d[p]), top));
    }
    return o;
  } else {
    unsigned u[8];
    corrected8<FF, SAFE>(v, f32, f64, u, vmax);
    Px8 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) o.d[p] = u[2 * p] | (u[2 * p + 1] << 16);
    return o;
  }
}

// Feather blend of one group that `cnt` (2 to 4) tiles touch, all with a usable flatfield plane
// and at least one of them covering all 8 pixels: loads of all sources issued together,
// float32 operations rounded one by one in placement order (two pixels per instruction).
// A tile that covers the group only partly contributes weight 0 outside its columns (adding
// 0 * value and 0 changes neither sum); its 16-byte load then runs into the neighbouring row
// of the same tile, which is why such a tile must not touch the group in its first or last
// row.  `ci` holds the candidate indices in placement order.
// In overwrite mode the same loads feed a selection instead: every pixel takes the value of
// the last placed tile that covers it (0 where none does), for 1 to 4 touching tiles.
template <int MODE, int FF, bool SAFE>
__device__ __forceinline__ Px8 blend_group(int y, int x, int cnt, const int (&ci)[4], const TileRef* list,
                                           const uint16_t* __restrict__ tiles, int H, int W,
                                           const void* __restrict__ flatfields, unsigned vmax) {
  constexpr size_t ffsize = FF == MISTITCH_FF_F64 ? 8 : 4;
  Px8 sv[4];
  F8 sf32[FF == MISTITCH_FF_F32 ? 4 : 1];
  D8 sf64[FF == MISTITCH_FF_F64 ? 4 : 1];
  int lys[4], lxs[4];
  const int64_t at = (int64_t)y * W + x;
  auto load = [&](int q) {
    const TileRef c = list[ci[q]];
    lys[q] = y - c.ty;
    lxs[q] = x - c.tx;
    sv[q] = *reinterpret_cast<const Px8*>(tiles + (c.src_base + at));
    if constexpr (FF == MISTITCH_FF_F32)
      sf32[q] = *reinterpret_cast<const F8*>(reinterpret_cast<const char*>(flatfields) + (c.ff_base + at) * (int64_t)ffsize);
    if constexpr (FF == MISTITCH_FF_F64)
      sf64[q] = *reinterpret_cast<const D8*>(reinterpret_cast<const char*>(flatfields) + (c.ff_base + at) * (int64_t)ffsize);
  };
  load(0);
  load(1);
  if (cnt > 2) {  // corners of the tile grid only; an absent fourth source re-reads the first
    load(2);
    load(3);
  }
  if constexpr (MODE == MISTITCH_FUSE_OVERWRITE) {
    unsigned out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = 0u;
    auto put = [&](int q) {
      unsigned u[8];
      corrected8<FF, SAFE>(sv[q], sf32[FF == MISTITCH_FF_F32 ? q : 0], sf64[FF == MISTITCH_FF_F64 ? q : 0], u, vmax);
#pragma unroll
      for (int i = 0; i < 8; ++i) out[i] = (unsigned)(lxs[q] + i) < (unsigned)W ? u[i] : out[i];
    };
    put(0);
    if (cnt > 1) put(1);
    if (cnt > 2) {
      put(2);
      if (cnt > 3) put(3);
    }
    Px8 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) o.d[p] = out[2 * p] | (out[2 * p + 1] << 16);
    return o;
  } else {
    f2 num[4], den[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) num[p] = den[p] = f2{0.0f, 0.0f};
    auto add = [&](int q) {
      const float wy = (float)min(lys[q] + 1, H - lys[q]);
      const float a = (float)(lxs[q] + 1), b = (float)(W - lxs[q]);  // ramp from the left / right tile edge
      unsigned u[8];
      corrected8<FF, SAFE>(sv[q], sf32[FF == MISTITCH_FF_F32 ? q : 0], sf64[FF == MISTITCH_FF_F64 ? q : 0], u, vmax);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f2 val = {(float)u[2 * p], (float)u[2 * p + 1]};
        const f2 up = f2{a, a} + f2{(float)(2 * p), (float)(2 * p + 1)};
        const f2 down = f2{b, b} - f2{(float)(2 * p), (float)(2 * p + 1)};
        // non-positive beyond either edge of the tile: weight 0 there
        const f2 w = f2{wy, wy} * f2{fmaxf(fminf(up.x, down.x), 0.0f), fmaxf(fminf(up.y, down.y), 0.0f)};
        num[p] = num[p] + w * val;
        den[p] = den[p] + w;
      }
    };
    add(0);
    add(1);
    if (cnt > 2) {
      add(2);
      if (cnt > 3) add(3);
    }
    Px8 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f2 t = div2_noscale(num[p], den[p]) + f2{0.5f, 0.5f};  // floor(t) by the truncating conversion: t > 0
      o.d[p] = pack2((unsigned int)t.x, (unsigned int)t.y);
    }
    return o;
  }
}

// Destination geometry of a mosaic that is born in chunk order (mistitch_fuse_chunked): plane p, row y, column x
// goes to element ((p * ncy + y / chunk_h) * ncx + x / chunk_w) * chunk_h * chunk_w + (y % chunk_h) * chunk_w +
// x % chunk_w, and the part of the last chunks beyond the mosaic (rows Hm .. ncy * chunk_h, columns Wm .. ncx * chunk_w)
// is written as zeros.  chunk_w is a multiple of 8 and the array 16-byte aligned, so an 8-pixel group that starts at a
// multiple of 8 is an aligned 16 bytes inside one chunk row.  chunk_h == 0: a plain row-major mosaic.
struct ChunkGeom {
  int chunk_h, chunk_w, ncy, ncx;
  __device__ __forceinline__ bool on() const { return chunk_h != 0; }
  __device__ __forceinline__ int Hp() const { return ncy * chunk_h; }
  __device__ __forceinline__ int Wp() const { return ncx * chunk_w; }
  __device__ __forceinline__ int64_t plane_elems() const { return (int64_t)ncy * ncx * chunk_h * chunk_w; }
  __device__ __forceinline__ int64_t row_base(int y) const {  // of row y inside its plane, chunk column 0
    return ((int64_t)(y / chunk_h) * ncx * chunk_h + y % chunk_h) * chunk_w;
  }
  __device__ __forceinline__ int col_off(int x) const { return (x / chunk_w) * chunk_h * chunk_w + x % chunk_w; }
};

// Work of one block.  1-D grid of 8 * gx * ceil(gy * n_planes / 8) workgroups with
// gx = ceil(ceil((Wm + 7) / 8) / (64 GX)) and gy = ceil((row1 - row0) / kBH).
// GX = 16-byte groups per lane along a row: a wave moves GX KiB of one mosaic row per step.
//
// Queue entries (one 32-bit word per group that is not a plain copy):
//   bits 0-7 group in the block row, 8-12 row in the block, 13-14 number of blended
//   sources less one (0: resolve pixel by pixel), 15-30 their candidate indices: two of 5
//   bits, or three or four of 4 bits, the last placed in the lowest bits.
// Blend entries fill the queue from the bottom, pixel-by-pixel entries from the top.
template <int MODE, int FF, int GX, bool SAFE, bool CH>
__device__ __forceinline__ void fuse_block(TileRef* list, unsigned* queue, unsigned* queue_len,
                                           const uint16_t* __restrict__ tiles, int n_tiles, int H, int W,
                                           int64_t tile_pitch, const int32_t* __restrict__ tile_yx,
                                           const int32_t* __restrict__ tile_plane,
                                           const void* __restrict__ flatfields,
                                           const int32_t* __restrict__ tile_ff, const int32_t* __restrict__ bins,
                                           uint16_t* __restrict__ mosaic, int Hm, int Wm, int row0, int row1, int gx,
                                           int gy, int n_planes, int64_t idle_off, int64_t idle_foff, const ChunkGeom cg,
                                           unsigned vmax) {
  constexpr int kQueue = kBH * 64 * GX * kWX;
  constexpr int kGroupBits = (64 * GX * kWX <= 256) ? 8 : (64 * GX * kWX <= 512 ? 9 : 10);  // group in the block row, then the row
  static_assert(64 * GX * kWX <= 1024 && kBH <= (1 << (13 - kGroupBits)), "13 bits of group and row");
  static_assert(GX <= 6, "a lane's copy sources are packed 5 bits per group into one word");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wx = wave % kWX, wy = wave / kWX;
  if (tid == 0) queue_len[0] = queue_len[1] = 0;
  // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  All
  // blocks of one block-row read the same flatfield rows (the tiles of a grid row
  // start at nearly the same y), so a whole block-row is given to ONE XCD: block-row
  // number = 8 * (j / gx) + (id % 8) with j = id / 8, and bx = j % gx.
  const int id = blockIdx.x;
  const int j = id >> 3;
  const int bx = j % gx;
  const int rid = ((j / gx) << 3) + (id & 7);  // plane * gy + by
  if (rid >= gy * n_planes) return;
  const int plane = rid / gy, by = rid - plane * gy;
  const int y0 = row0 + by * kBH, y1 = min(y0 + kBH, row1);
  // the tiles touching this block, in placement order, were listed by k_fuse_bins
  const int32_t* bin = bins + ((int64_t)rid * gx + bx) * (kMaxCand + 1);
  const int total = bin[0];
  bool has_ff = false;
  if (tid < total && tid < kMaxCand) {
    const TileRef c = make_ref(bin[1 + tid], tile_yx, tile_ff, H, W, tile_pitch);
    list[tid] = c;
    has_ff = c.ff >= 0;
  }
  // candidates that a plain copy or a batched blend can take: with a flatfield kernel, those that have a plane
  const unsigned okmask = FF == MISTITCH_FF_NONE ? 0xffffffffu : (unsigned)__ballot(has_ff && wave == 0);
  if (tid == 0) queue_len[2] = okmask;
  __syncthreads();
  const unsigned ffok = queue_len[2];
  // more tiles than the list holds, or tiles or a mosaic narrower than a group: every group is resolved pixel by pixel
  const bool overflow = total > kMaxCand;
  const bool generic_only = overflow || W < 8 || Wm < 8;
  const int ncand = overflow ? n_tiles : total;
  constexpr bool chunked = CH;  // (cg.on())
  uint16_t* mplane = mosaic + (chunked ? (int64_t)plane * cg.plane_elems() : (int64_t)plane * Hm * Wm);
  const int Wp = chunked ? cg.Wp() : Wm;  // columns that are written
  const int w_idx = (bx * kWX + wx) * (64 * GX) + lane;  // first aligned group of this lane within a row
  int col_off[GX];  // chunk order: where the lane's groups lie inside a chunk row (they start at multiples of 8)
#pragma unroll
  for (int r = 0; r < GX; ++r) col_off[r] = chunked ? cg.col_off(min(8 * w_idx + 512 * r, Wp - 8)) : 0;
  constexpr size_t ffsize = FF == MISTITCH_FF_F64 ? 8 : 4;

  // One mosaic row per wave and step.  The tiles whose rows include y are found once
  // per row (one candidate per lane, a ballot); a lane then classifies its GX groups
  // against those few without touching pixel memory (0: uncovered, 1: plain copy from
  // one tile, 2: queued, 3: outside the mosaic), issues all copy loads back to back
  // and stores.
  //
  // The classification of a row depends on the row only through (a) the alignment of its first pixel (its address
  // modulo 16 bytes, which repeats every 8 rows whatever the mosaic's width), (b) the set of tiles whose rows
  // include it and (c) whether it is the first or last row of one of them.  A wave therefore takes its rows in the
  // order 0, 8, 16, 24, 4, 12, 20, 28 (+ its number): consecutive rows share (a), and away from the tiles' first
  // and last rows (b) and (c) too, so that the classification of the previous row stands and only addresses,
  // loads and stores remain (vector instructions issue on 59 % of this kernel's cycles: they, not memory, set its pace).
  constexpr int kWaves = kWY;  // waves that share a block's rows
  constexpr int kRowsPerWave = (kBH + kWaves - 1) / kWaves;
  unsigned kinds = 0;        // two bits per group
  unsigned csel = 0;         // 5 bits per group: the list entry a plain copy reads
  unsigned qinfo[GX];        // queue entry of the group less its row
  unsigned qbatched = 0;     // bit r: the entry goes to the batched blends
  int prev_a = -1;
  unsigned prev_ymask = 0, prev_edges = 1;  // no previous row yet
#pragma unroll
  for (int r = 0; r < GX; ++r) qinfo[r] = 0;
  for (int it = 0; it < kRowsPerWave; ++it) {
    const int half = (kRowsPerWave + 1) / 2;
    const int kk = it < half ? 2 * it : 2 * (it - half) + 1;  // even steps first, then the odd ones
    const int y = y0 + wy + kk * kWaves;
    if (y >= y1) continue;  // wave-uniform
    uint16_t* rowp = mplane + (chunked ? cg.row_base(y) : (int64_t)y * Wm);
    const int a = chunked ? 0 : (int)((reinterpret_cast<uintptr_t>(rowp) >> 1) & 7);
    const int64_t yW = (int64_t)y * W;
    if (chunked && y >= Hm) {  // (chunk order) a row of the last chunks below the mosaic: zeros, whatever tiles stick out that far
#pragma unroll
      for (int r = 0; r < GX; ++r) {
        const int x = 8 * w_idx + 512 * r;
        if (x < Wp) store_px8_stream(rowp + col_off[r], Px8{{0u, 0u, 0u, 0u}});
      }
      continue;  // wave-uniform; the cached classification still describes the last mosaic row this wave took
    }
    unsigned ymask = 0, edges = 0;
    if (!generic_only) {
      const int ly_mine = y - list[lane & (kMaxCand - 1)].ty;
      const bool in_rows = lane < ncand && (unsigned)ly_mine < (unsigned)H;
      ymask = (unsigned)__ballot(in_rows);
      edges = (unsigned)__ballot(in_rows && ((ly_mine == 0) | (ly_mine == H - 1)));
    }
    const int x0 = 8 * w_idx - a;
    if (!(a == prev_a && ymask == prev_ymask && (edges | prev_edges) == 0)) {  // wave-uniform
      int cnt[GX];
      unsigned idx[GX];    // touching candidates, 5 bits each, the last placed in the lowest bits
      bool bad[GX];        // some touching tile covers the group only partly or lacks a flatfield plane
      bool last_bad[GX];   // the same for the last placed touching tile alone
      bool hard[GX];       // a touching tile that no batched blend or selection can take (see blend_group)
      bool full[GX];       // some touching tile covers all 8 pixels
#pragma unroll
      for (int r = 0; r < GX; ++r) {
        cnt[r] = 0;
        idx[r] = 0;
        bad[r] = last_bad[r] = hard[r] = full[r] = false;
      }
      for (unsigned m = ymask; m; m &= m - 1u) {  // wave-uniform loop over the row's tiles
        const int ci = __builtin_ctz(m);
        const int tx = list[ci].tx;
        const bool not_ok = !((ffok >> ci) & 1u);
        const bool edge_row = (edges >> ci) & 1u;  // a partly covering tile's load would leave the tile here
#pragma unroll
        for (int r = 0; r < GX; ++r) {
          const int lx0 = x0 + 512 * r - tx;
          const bool touch = (unsigned)(lx0 + 7) < (unsigned)(W + 7);
          const bool cut = (unsigned)lx0 > (unsigned)(W - 8);
          const bool part = cut | not_ok;
          idx[r] = touch ? ((idx[r] << 5) | (unsigned)ci) : idx[r];
          cnt[r] += touch ? 1 : 0;
          bad[r] = bad[r] | (touch & part);
          last_bad[r] = touch ? part : last_bad[r];
          hard[r] = hard[r] | (touch & (not_ok | (cut & edge_row)));
          full[r] = full[r] | (touch & !cut);
        }
      }
      kinds = csel = qbatched = 0;
#pragma unroll
      for (int r = 0; r < GX; ++r) {
        const int x = x0 + 512 * r;
        const bool outside = !((unsigned)(x + 7) < (unsigned)(Wp + 7));
        const bool inside = (unsigned)x <= (unsigned)(Wm - 8);  // not cut by the mosaic edge
        const bool copy = MODE == MISTITCH_FUSE_OVERWRITE ? !last_bad[r] : (cnt[r] == 1) & !bad[r];
        int k = (cnt[r] == 0) ? 0 : (copy ? 1 : 2);
        k = (!inside | generic_only) ? 2 : k;
        k = x >= Wm ? 0 : k;  // (chunk order) the columns of the last chunks beyond the mosaic: zeros
        k = outside ? 3 : k;
        kinds |= (unsigned)k << (2 * r);
        csel |= (idx[r] & 31u) << (5 * r);
        // the queue entry of a group that is not a plain copy: two sources may have any list index; three or four
        // must all lie below 16 (4-bit fields)
        const unsigned i5 = idx[r];
        const bool small = ((i5 | (i5 >> 5) | (i5 >> 10) | (i5 >> 15)) & 16u) == 0;
        const bool batched = !generic_only && inside && !hard[r] && (cnt[r] <= 2 || (cnt[r] <= 4 && small)) &&
                             (MODE == MISTITCH_FUSE_FEATHER ? (full[r] && cnt[r] >= 2) : cnt[r] >= 1);
        const unsigned packed = cnt[r] <= 2 ? (i5 & 0x3ffu)
                                            : ((i5 & 15u) | ((i5 >> 1) & 0xf0u) | ((i5 >> 2) & 0xf00u) | ((i5 >> 3) & 0xf000u));
        qinfo[r] = (unsigned)(wx * 64 * GX + lane + 64 * r) | (batched ? ((unsigned)(cnt[r] - 1) << 13) | (packed << 15) : 0u);
        qbatched |= (batched ? 1u : 0u) << r;
      }
    }
    prev_a = a;
    prev_ymask = ymask;
    prev_edges = edges;
    Px8 v[GX];
    F8 f32[FF == MISTITCH_FF_F32 ? GX : 1];
    D8 f64[FF == MISTITCH_FF_F64 ? GX : 1];
#pragma unroll
    for (int r = 0; r < GX; ++r) {
      const int x = x0 + 512 * r;
      const unsigned k = (kinds >> (2 * r)) & 3u;
      // copy loads never sit behind a branch: lanes without a plain copy read tile 0's first pixels (idle_off = 0;
      // the host points both idle offsets at the workspace when the tile or flatfield array holds fewer than 8 values)
      const TileRef* c = list + ((csel >> (5 * r)) & 31u);
      const int64_t at = yW + x;
#if defined(MS_FUSE_ABLATE) && MS_FUSE_ABLATE == 4  /* measurement build: every pixel load of a copy starts on a 4-byte boundary */
      const int64_t off = (k == 1 ? c->src_base + at : idle_off) & ~(int64_t)1;
#elif defined(MS_FUSE_ABLATE) && MS_FUSE_ABLATE == 5  /* measurement build: ... on a 16-byte boundary */
      const int64_t off = (k == 1 ? c->src_base + at : idle_off) & ~(int64_t)7;
#else
      const int64_t off = k == 1 ? c->src_base + at : idle_off;
#endif
#if defined(MS_FUSE_ABLATE) && MS_FUSE_ABLATE == 3  /* measurement build: every flatfield load of a copy falls into one 64 KiB window */
      const int64_t foff = (k == 1 ? c->ff_base + at : idle_foff) & 0x3fff;
#elif defined(MS_FUSE_ABLATE) && MS_FUSE_ABLATE == 6  /* measurement build: every flatfield load of a copy starts on a 32-byte boundary */
      const int64_t foff = (k == 1 ? c->ff_base + at : idle_foff) & ~(int64_t)7;
#else
      const int64_t foff = k == 1 ? c->ff_base + at : idle_foff;
#endif
      v[r] = *reinterpret_cast<const Px8*>(tiles + off);
      if constexpr (FF == MISTITCH_FF_F32)
        f32[r] = *reinterpret_cast<const F8*>(reinterpret_cast<const char*>(flatfields) + foff * (int64_t)ffsize);
      if constexpr (FF == MISTITCH_FF_F64)
        f64[r] = *reinterpret_cast<const D8*>(reinterpret_cast<const char*>(flatfields) + foff * (int64_t)ffsize);
    }
#pragma unroll
    for (int r = 0; r < GX; ++r) {
      MS_PIN_PX8(v[r]);
      if constexpr (FF == MISTITCH_FF_F32) MS_PIN_8(f32[r]);
      if constexpr (FF == MISTITCH_FF_F64) MS_PIN_8(f64[r]);
    }
#pragma unroll
    for (int r = 0; r < GX; ++r) {
      const unsigned k = (kinds >> (2 * r)) & 3u;
      const int x = x0 + 512 * r;
      uint16_t* dst = rowp + (chunked ? col_off[r] : x);
      if (k == 1) {
        if constexpr (FF == MISTITCH_FF_NONE)
          store_px8_stream(dst, v[r]);
        else
          store_px8_stream(dst, corrected8_packed<FF, SAFE>(v[r], f32[FF == MISTITCH_FF_F32 ? r : 0],
                                                            f64[FF == MISTITCH_FF_F64 ? r : 0], vmax));
      } else if (k == 0) {
        store_px8_stream(dst, Px8{{0u, 0u, 0u, 0u}});
      } else if (k == 2) {
        // queued for the whole workgroup: these groups cluster in the overlap bands, and
        // handled in place they would stall every wave that owns a few
        const unsigned e = qinfo[r] | ((unsigned)(y - y0) << kGroupBits);
        if ((qbatched >> r) & 1u)
          queue[atomicAdd(&queue_len[0], 1u)] = e;
        else
          queue[kQueue - 1 - atomicAdd(&queue_len[1], 1u)] = e;
      }
    }
  }
  __syncthreads();
  {
    const unsigned n_blend = queue_len[0];
    for (unsigned i = tid; i < n_blend; i += kFuseThreads) {
      const unsigned e = queue[i];
      const int y = y0 + (int)((e >> kGroupBits) & ((1u << (13 - kGroupBits)) - 1u));
      uint16_t* rowp = mplane + (chunked ? cg.row_base(y) : (int64_t)y * Wm);
      const int a = chunked ? 0 : (int)((reinterpret_cast<uintptr_t>(rowp) >> 1) & 7);
      const int x = 8 * (bx * (64 * GX * kWX) + (int)(e & ((1u << kGroupBits) - 1u))) - a;
      const int cnt = (int)((e >> 13) & 3u) + 1;
      const unsigned f = e >> 15;
      int ci[4];  // placement order; absent sources repeat the first
      if (cnt <= 2) {
        ci[1] = (int)(f & 31u);
        ci[0] = cnt == 2 ? (int)((f >> 5) & 31u) : ci[1];
        ci[2] = ci[3] = ci[0];
      } else {
        ci[0] = (int)((f >> (4 * (cnt - 1))) & 15u);
        ci[1] = (int)((f >> (4 * (cnt - 2))) & 15u);
        ci[2] = (int)((f >> (4 * (cnt - 3))) & 15u);
        ci[3] = cnt == 4 ? (int)(f & 15u) : ci[0];
      }
      store_px8_stream(rowp + (chunked ? cg.col_off(x) : x), blend_group<MODE, FF, SAFE>(y, x, cnt, ci, list, tiles, H, W, flatfields, vmax));
    }
  }
  const unsigned n_generic = queue_len[1];
  for (unsigned i = tid; i < n_generic; i += kFuseThreads) {
    const unsigned e = queue[kQueue - 1 - i];
    const int y = y0 + (int)((e >> kGroupBits) & ((1u << (13 - kGroupBits)) - 1u));
    uint16_t* rowp = mplane + (chunked ? cg.row_base(y) : (int64_t)y * Wm);
    const int a = chunked ? 0 : (int)((reinterpret_cast<uintptr_t>(rowp) >> 1) & 7);
    const int x = 8 * (bx * (64 * GX * kWX) + (int)(e & ((1u << kGroupBits) - 1u))) - a;
    const unsigned vm = (((1u << min(8, Wm - x)) - 1u) & ~((1u << max(0, -x)) - 1u)) & 0xffu;
    uint16_t px[8];
    resolve_window<MODE, FF>(y, x, vm, list, ncand, overflow, plane, tiles, H, W, tile_pitch, tile_yx, tile_plane,
                             flatfields, tile_ff, px, vmax);
    // chunk order: the whole group is written, zeros (as resolve_window leaves them) beyond the mosaic's last column
    store_window(rowp + (chunked ? cg.col_off(x) : x), px, chunked ? 0xffu : vm);
  }
}

// `ff_safe` points at the verdict of k_ff_check (1: every flatfield value used by this
// call admits the scaling-free division).
template <int MODE, int FF, int GX, bool CH>
__global__ __launch_bounds__(kFuseThreads) void k_fuse(const uint16_t* __restrict__ tiles, int n_tiles, int H,
                                                       int W, int64_t tile_pitch,
                                                       const int32_t* __restrict__ tile_yx,
                                                       const int32_t* __restrict__ tile_plane,
                                                       const void* __restrict__ flatfields,
                                                       const int32_t* __restrict__ tile_ff,
                                                       const int32_t* __restrict__ bins,
                                                       const int32_t* __restrict__ ff_safe,
                                                       uint16_t* __restrict__ mosaic, int Hm, int Wm, int row0,
                                                       int row1, int gx, int gy, int n_planes, int64_t idle_off,
                                                       int64_t idle_foff, const ChunkGeom cg, unsigned vmax) {
  __shared__ TileRef list[kMaxCand];
  __shared__ unsigned queue[kBH * 64 * GX * kWX];
  __shared__ unsigned queue_len[3];  // blend entries, pixel-by-pixel entries, candidates with a flatfield plane
  if constexpr (FF == MISTITCH_FF_F32) {
    if (*ff_safe) {
      fuse_block<MODE, FF, GX, true, CH>(list, queue, queue_len, tiles, n_tiles, H, W, tile_pitch, tile_yx, tile_plane,
                                     flatfields, tile_ff, bins, mosaic, Hm, Wm, row0, row1, gx, gy, n_planes, idle_off,
                                     idle_foff, cg, vmax);
      return;
    }
  }
  fuse_block<MODE, FF, GX, false, CH>(list, queue, queue_len, tiles, n_tiles, H, W, tile_pitch, tile_yx, tile_plane,
                                  flatfields, tile_ff, bins, mosaic, Hm, Wm, row0, row1, gx, gy, n_planes, idle_off, idle_foff, cg, vmax);
}

// Are all float32 flatfield values that this call uses finite and within [2^-16, 2^16]?
// grid (kFfChunks, kFfPlanes): block (c, p) scans chunk c of plane p if a tile uses the plane
// (`many_planes`: bit p of verdict[1..2], written by k_fuse_bins before; else the single plane 0);
// a violation clears verdict[0], which has been set before (k_fuse_bins, or a memset).
constexpr int kFfChunks = 512, kFfPlanes = 64;
static_assert(kFfPlanes == 64, "k_fuse_bins keeps one bit per plane in two 32-bit words");
__global__ __launch_bounds__(256) void k_ff_check(const float* __restrict__ flatfields, int many_planes,
                                                  int64_t plane_px, int32_t* verdict) {
  const int p = blockIdx.y;
  if (many_planes ? !(((unsigned)verdict[1 + (p >> 5)] >> (p & 31)) & 1u) : p != 0) return;
  const float* plane = flatfields + (int64_t)p * plane_px;
  const int64_t per = ((plane_px + kFfChunks - 1) / kFfChunks + 3) & ~(int64_t)3;
  const int64_t i0 = per * blockIdx.x, i1 = min(plane_px, i0 + per);
  int bad = 0;
  // positive, exponent within +-16: bits in [0x37800000, 0x47800000]
  auto worst_of = [](const F8& q) {  // 4-byte aligned 16-byte read of the first half
    return max(max(__float_as_uint(q.v[0]) - 0x37800000u, __float_as_uint(q.v[1]) - 0x37800000u),
               max(__float_as_uint(q.v[2]) - 0x37800000u, __float_as_uint(q.v[3]) - 0x37800000u));
  };
  const int64_t stride = 4 * (int64_t)blockDim.x;
  int64_t i = i0 + 4 * threadIdx.x;
  for (; i + 3 * stride + 4 <= i1; i += 4 * stride) {  // four reads in flight
    const unsigned w0 = worst_of(*reinterpret_cast<const F8*>(plane + i));
    const unsigned w1 = worst_of(*reinterpret_cast<const F8*>(plane + i + stride));
    const unsigned w2 = worst_of(*reinterpret_cast<const F8*>(plane + i + 2 * stride));
    const unsigned w3 = worst_of(*reinterpret_cast<const F8*>(plane + i + 3 * stride));
    bad |= max(max(w0, w1), max(w2, w3)) > 0x10000000u;
  }
  for (; i < i1; i += stride) {
    if (i + 4 <= i1) {
      bad |= worst_of(*reinterpret_cast<const F8*>(plane + i)) > 0x10000000u;
    } else {
      for (int64_t e = i; e < i1; ++e) bad |= !(__float_as_uint(plane[e]) - 0x37800000u <= 0x10000000u);
    }
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) *verdict = 0;
}

// out = trunc(clip(in / ff)); one flatfield plane shared by n_tiles images.  One
// 8-pixel group per thread, bpt blocks per tile; the tail (n_px_per_tile % 8) is
// handled by k_flatfield_tail.  `ff_safe`: verdict of k_ff_check for the plane.
template <int FF>
__global__ __launch_bounds__(256) void k_flatfield(const uint16_t* in, const void* __restrict__ ff, uint16_t* out,
                                                   int64_t n_px_per_tile, unsigned bpt, unsigned tile0,
                                                   const int32_t* __restrict__ ff_safe, unsigned vmax) {
  const unsigned t = tile0 + blockIdx.x / bpt;
  const int64_t q = (int64_t)(blockIdx.x % bpt) * 256 + threadIdx.x;
  if (q >= n_px_per_tile / 8) return;
  const Px8 v = *reinterpret_cast<const Px8*>(in + t * n_px_per_tile + q * 8);
  F8 f32{};
  D8 f64{};
  if constexpr (FF == MISTITCH_FF_F32) f32 = *reinterpret_cast<const F8*>(reinterpret_cast<const float*>(ff) + q * 8);
  if constexpr (FF == MISTITCH_FF_F64) f64 = *reinterpret_cast<const D8*>(reinterpret_cast<const double*>(ff) + q * 8);
  Px8 o;
  if (FF == MISTITCH_FF_F32 && *ff_safe)
    o = corrected8_packed<FF, FF == MISTITCH_FF_F32>(v, f32, f64, vmax);
  else
    o = corrected8_packed<FF, false>(v, f32, f64, vmax);
  *reinterpret_cast<Px8*>(out + t * n_px_per_tile + q * 8) = o;
}

template <int FF>
__global__ void k_flatfield_tail(const uint16_t* in, const void* ff, uint16_t* out, int64_t n_px_per_tile,
                                 int n_tiles, unsigned vmax) {
  const int64_t tail0 = n_px_per_tile / 8 * 8;
  const int ntail = (int)(n_px_per_tile - tail0);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntail * n_tiles) return;
  const int t = i / ntail, e = i - t * ntail;
  const char* ffrow = reinterpret_cast<const char*>(ff) + tail0 * (FF == MISTITCH_FF_F64 ? 8 : 4);
  out[t * n_px_per_tile + tail0 + e] = flatfield_one<FF>(in[t * n_px_per_tile + tail0 + e], ffrow, e, vmax);
}

// Per-pixel maximum over n images.
__global__ __launch_bounds__(256) void k_mip(const uint16_t* __restrict__ in, int n, int64_t pitch, int64_t n_pixels,
                                             uint16_t* __restrict__ out) {
  const int64_t ngroups = n_pixels / 8;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
    Px8 best = *reinterpret_cast<const Px8*>(in + g * 8);
    for (int z = 1; z < n; ++z) {
      const Px8 v = *reinterpret_cast<const Px8*>(in + z * pitch + g * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) best.set(i, max(v.at(i), best.at(i)));
    }
    *reinterpret_cast<Px8*>(out + g * 8) = best;
  }
  if (blockIdx.x == 0) {
    for (int64_t e = ngroups * 8 + threadIdx.x; e < n_pixels; e += blockDim.x) {
      uint16_t best = in[e];
      for (int z = 1; z < n; ++z) best = in[z * pitch + e] > best ? in[z * pitch + e] : best;
      out[e] = best;
    }
  }
}

// Resolution pyramid: level l holds the mean of every 2^l x 2^l block of the plane
// (blocks that would stick out of the image are dropped), as float64 - what
// dask.array.coarsen(np.mean, image, {y: 2^l, x: 2^l}, trim_excess=True) yields for each
// level in Stitcher.generate_pyramid (stitcher.py:652-668).  All levels come out of ONE
// read of the image: a workgroup owns a 64 x 256 pixel tile, forms the 2 x 2 sums in
// registers and the coarser ones through LDS; the sums are exact integers, the division by
// 4^l exact in float64.
constexpr int kPyrTileH = 64, kPyrTileW = 256, kPyrMaxLevels = 7;
struct PyrLevels {
  double* out[kPyrMaxLevels];  // [l]: level l (index 0 unused)
  int och, ocw;                // chunk shape of the LEVELS when they are written in chunk order (0: row-major planes)
  // where pixel (oy, ox) of plane `plane` of a level of Hl x Wl pixels goes
  __device__ __forceinline__ int64_t at(int plane, int Hl, int Wl, int oy, int ox) const {
    if (och == 0) return ((int64_t)plane * Hl + oy) * Wl + ox;
    const int ncy = (Hl + och - 1) / och, ncx = (Wl + ocw - 1) / ocw;
    return (((int64_t)plane * ncy + oy / och) * ncx + ox / ocw) * ((int64_t)och * ocw) + (int64_t)(oy % och) * ocw + ox % ocw;
  }
};

// Zeros in what sticks out of a level written in chunk order: the rows below the level and the columns right of it,
// up to whole chunks.  grid (blocks, levels - 1, planes), grid-stride over the padding elements of a plane.
__global__ __launch_bounds__(256) void k_pyramid_padding(int H, int W, PyrLevels lv) {
  const int l = 1 + blockIdx.y, plane = blockIdx.z;
  const int Hl = H >> l, Wl = W >> l;
  if (Hl == 0 || Wl == 0 || lv.out[l] == nullptr) return;
  const int Hp = (Hl + lv.och - 1) / lv.och * lv.och, Wp = (Wl + lv.ocw - 1) / lv.ocw * lv.ocw;
  const int64_t right = (int64_t)Hl * (Wp - Wl), below = (int64_t)(Hp - Hl) * Wp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < right + below; e += (int64_t)gridDim.x * blockDim.x) {
    int oy, ox;
    if (e < right) {
      oy = (int)(e / (Wp - Wl));
      ox = Wl + (int)(e % (Wp - Wl));
    } else {
      const int64_t f = e - right;
      oy = Hl + (int)(f / Wp);
      ox = (int)(f % Wp);
    }
    lv.out[l][lv.at(plane, Hl, Wl, oy, ox)] = 0.0;
  }
}

// `cg`: the image is an array of chunks (ChunkGeom, chunk_w a multiple of 8) instead of row-major planes.
__global__ __launch_bounds__(256) void k_pyramid(const uint16_t* __restrict__ image, int H, int W, int n_levels,
                                                 PyrLevels lv, const ChunkGeom cg) {
  __shared__ unsigned sums[2][kPyrTileH / 2][kPyrTileW / 2 + 1];  // ping-pong block sums of the current level
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const int y0 = blockIdx.y * kPyrTileH, x0 = blockIdx.x * kPyrTileW;
  const bool chunked = cg.on();
  const uint16_t* img = image + (chunked ? (int64_t)plane * cg.plane_elems() : (int64_t)plane * H * W);
  auto at = [&](int yy, int xx) -> const uint16_t* {  // pixel (yy, xx) of the plane; xx a multiple of 8 stays inside a chunk row
    return chunked ? img + cg.row_base(yy) + cg.col_off(xx) : img + (int64_t)yy * W + xx;
  };
  // level 1: thread (j, g) sums rows 2j, 2j+1 of the 8-pixel group g; 32 x 32 such cells per tile, 4 per thread
  for (int cell = tid; cell < (kPyrTileH / 2) * (kPyrTileW / 8); cell += 256) {
    const int j = cell / (kPyrTileW / 8), g = cell - j * (kPyrTileW / 8);
    const int y = y0 + 2 * j, x = x0 + 8 * g;
    unsigned px[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (y + r < H && x + 8 <= W) {
        const Px8 v = *reinterpret_cast<const Px8*>(at(y + r, x));
#pragma unroll
        for (int i = 0; i < 8; ++i) px[r][i] = v.at(i);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) px[r][i] = (y + r < H && x + i < W) ? at(y + r, x)[i] : 0u;
      }
    }
    const int H1 = H >> 1, W1 = W >> 1;
    const int oy = y >> 1, ox = x >> 1;
    // (chunk order) the four values of a cell lie side by side unless a chunk ends among them
    const bool one_run = lv.och == 0 || ox % lv.ocw + 3 < lv.ocw;
    const int64_t first = (n_levels > 1 && oy < H1 && ox < W1) ? lv.at(plane, H1, W1, oy, ox) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned sum = px[0][2 * i] + px[0][2 * i + 1] + px[1][2 * i] + px[1][2 * i + 1];
      sums[0][j][4 * g + i] = sum;
      if (n_levels > 1 && oy < H1 && ox + i < W1)
        lv.out[1][one_run ? first + i : lv.at(plane, H1, W1, oy, ox + i)] = (double)sum * 0.25;
    }
  }
  int cur = 0, th = kPyrTileH / 2, tw = kPyrTileW / 2;  // extent of the sums held in sums[cur]
  for (int l = 2; l < n_levels; ++l) {
    __syncthreads();
    const int nh = th / 2, nw = tw / 2;
    const int Hl = H >> l, Wl = W >> l;
    const double scale = 1.0 / (double)(1u << (2 * l));
    for (int cell = tid; cell < nh * nw; cell += 256) {
      const int j = cell / nw, i = cell - j * nw;
      const unsigned sum = sums[cur][2 * j][2 * i] + sums[cur][2 * j][2 * i + 1] + sums[cur][2 * j + 1][2 * i] +
                           sums[cur][2 * j + 1][2 * i + 1];
      sums[cur ^ 1][j][i] = sum;
      const int oy = (y0 >> l) + j, ox = (x0 >> l) + i;
      if (oy < Hl && ox < Wl) lv.out[l][lv.at(plane, Hl, Wl, oy, ox)] = (double)sum * scale;
    }
    cur ^= 1;
    th = nh;
    tw = nw;
  }
}

template <int MODE, int FF, int GX>
static void launch_fuse(mistitch_ctx* ctx, dim3 grid, const uint16_t* tiles, int n_tiles, int H, int W, int64_t pitch,
                        const int32_t* yx, const int32_t* plane, const void* ffs, const int32_t* tff,
                        const int32_t* bins, const int32_t* ff_safe, uint16_t* mosaic, int Hm, int Wm, int row0,
                        int row1, int64_t idle_off, int64_t idle_foff, const ChunkGeom& cg) {
  const int gx = (int)grid.x, gy = (int)grid.y, n_planes = (int)grid.z;
  const unsigned n_wg = 8u * gx * (unsigned)((gy * n_planes + 7) / 8);
  if (cg.chunk_h != 0)
    MS_LAUNCH(ctx, MODE == MISTITCH_FUSE_OVERWRITE ? "k_fuse_overwrite" : "k_fuse_feather", (k_fuse<MODE, FF, GX, true>),
              dim3(n_wg), dim3(kFuseThreads), 0, tiles, n_tiles, H, W, pitch, yx, plane, ffs, tff, bins, ff_safe, mosaic,
              Hm, Wm, row0, row1, gx, gy, n_planes, idle_off, idle_foff, cg, (unsigned)ctx->pixel_max);
  else
    MS_LAUNCH(ctx, MODE == MISTITCH_FUSE_OVERWRITE ? "k_fuse_overwrite" : "k_fuse_feather", (k_fuse<MODE, FF, GX, false>),
              dim3(n_wg), dim3(kFuseThreads), 0, tiles, n_tiles, H, W, pitch, yx, plane, ffs, tff, bins, ff_safe, mosaic,
              Hm, Wm, row0, row1, gx, gy, n_planes, idle_off, idle_foff, cg, (unsigned)ctx->pixel_max);
}

}  // namespace mistitch

using namespace mistitch;

// Workspace of the allocation-free entry point: the per-block tile lists of ONE fusion call.  Every GPU
// the library runs on holds its own copy of the array at its own address (hipGetSymbolAddress with that
// device current); host code never passes the array's name to a kernel, only those addresses.
constexpr size_t kStaticScratchBytes = 16u << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
static void* g_scratch_of_device[kMaxDevices];  // filled by fuse_static_setup, one entry per device

namespace mistitch {
// Called by mistitch_create with `device` current: looks up this device's copy of `scratch` once.
int fuse_static_setup(int device) {
  MS_REQUIRE(device >= 0 && device < kMaxDevices, MISTITCH_ERR_UNSUPPORTED, "device index %d beyond %d", device,
             kMaxDevices - 1);
  if (__atomic_load_n(&g_scratch_of_device[device], __ATOMIC_ACQUIRE) != nullptr) return MISTITCH_OK;
  void* p = nullptr;
  MS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  MS_REQUIRE(p != nullptr, MISTITCH_ERR_HIP, "hipGetSymbolAddress returned no address for device %d", device);
  __atomic_store_n(&g_scratch_of_device[device], p, __ATOMIC_RELEASE);  // every caller computes the same value
  return MISTITCH_OK;
}
}  // namespace mistitch

// Body of mistitch_fuse / mistitch_fuse_static.  `static_ws`: this device's copy of `scratch`, or NULL to
// take (and if need be grow) the context's own buffer.
static int fuse_impl(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w, int64_t tile_pitch,
                     const int32_t* tile_yx, const int32_t* tile_plane, const void* flatfields, int ff_type,
                     const int32_t* tile_ff, uint16_t* mosaic, int n_planes, int mosaic_h, int mosaic_w, int row0,
                     int row1, int mode, void* static_ws, int chunk_h = 0, int chunk_w = 0) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(mosaic, MISTITCH_ERR_ARG, "mosaic is NULL");
  MS_REQUIRE(n_tiles >= 0 && n_planes > 0 && mosaic_h > 0 && mosaic_w > 0, MISTITCH_ERR_ARG, "bad mosaic geometry");
  MS_REQUIRE(n_tiles == 0 || (tiles && tile_yx && tile_plane), MISTITCH_ERR_ARG, "NULL tile array");
  MS_REQUIRE(tile_h > 0 && tile_w > 0 && tile_pitch >= (int64_t)tile_h * tile_w, MISTITCH_ERR_ARG, "bad tile geometry");
  MS_REQUIRE(0 <= row0 && row0 <= row1 && row1 <= mosaic_h, MISTITCH_ERR_ARG, "row band [%d, %d) outside mosaic", row0,
             row1);
  MS_REQUIRE(mode == MISTITCH_FUSE_OVERWRITE || mode == MISTITCH_FUSE_FEATHER, MISTITCH_ERR_ARG, "unknown mode %d", mode);
  MS_REQUIRE(ff_type == MISTITCH_FF_NONE || ff_type == MISTITCH_FF_F32 || ff_type == MISTITCH_FF_F64, MISTITCH_ERR_ARG,
             "unknown flatfield type %d", ff_type);
  if (ff_type != MISTITCH_FF_NONE && n_tiles > 0)
    MS_REQUIRE(flatfields && tile_ff, MISTITCH_ERR_ARG, "flatfield type given without flatfields / tile_ff");
  if (row1 == row0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  // chunk order: the mosaic is an array of whole chunks; rows below the mosaic in the last chunks belong to the band that
  // holds the mosaic's last row, columns beyond its last column to every band
  ChunkGeom cg{0, 0, 0, 0};
  int out_w = mosaic_w, row_end = row1;
  if (chunk_h != 0) {
    MS_REQUIRE(chunk_h > 0 && chunk_w > 0 && chunk_w % 8 == 0, MISTITCH_ERR_UNSUPPORTED,
               "chunks of %d x %d: the chunk width must be a multiple of 8 pixels (lay the mosaic out with mistitch_chunk_layout instead)",
               chunk_h, chunk_w);
    MS_REQUIRE((reinterpret_cast<uintptr_t>(mosaic) & 15u) == 0, MISTITCH_ERR_ARG, "the chunk array is not 16-byte aligned");
    cg = ChunkGeom{chunk_h, chunk_w, (mosaic_h + chunk_h - 1) / chunk_h, (mosaic_w + chunk_w - 1) / chunk_w};
    MS_REQUIRE((int64_t)cg.ncx * chunk_h * chunk_w < (1ll << 31) && (int64_t)cg.ncy * chunk_h < (1ll << 31) &&
                   (int64_t)cg.ncx * chunk_w < (1ll << 31),
               MISTITCH_ERR_UNSUPPORTED, "a row of chunks of 2^31 elements or more");
    out_w = cg.ncx * chunk_w;
    if (row1 == mosaic_h) row_end = cg.ncy * chunk_h;
  }
  if (n_tiles == 0) {
    // no tile reaches the band: it is what the reference's zero-initialised output array holds (stitcher.py:154-181);
    // the fusion kernels, whose idle lanes read the first tile's pixels, are not launched on an empty tile array
    if (chunk_h != 0) {  // the rows of the band inside every chunk of a chunk row are contiguous, chunk by chunk
      const size_t chunk_bytes = (size_t)chunk_h * chunk_w * sizeof(uint16_t);
      for (int cy = row0 / chunk_h; cy * chunk_h < row_end; ++cy) {
        const int ra = std::max(row0, cy * chunk_h) - cy * chunk_h, rb = std::min(row_end, (cy + 1) * chunk_h) - cy * chunk_h;
        for (int pl = 0; pl < n_planes; ++pl)
          MS_HIP(hipMemset2DAsync(mosaic + (((int64_t)pl * cg.ncy + cy) * cg.ncx * chunk_h + ra) * chunk_w, chunk_bytes, 0,
                                  (size_t)(rb - ra) * chunk_w * sizeof(uint16_t), (size_t)cg.ncx, ctx->stream));
      }
      return MISTITCH_OK;
    }
    MS_HIP(hipMemset2DAsync(mosaic + (int64_t)row0 * mosaic_w, (size_t)mosaic_h * mosaic_w * sizeof(uint16_t), 0,
                            (size_t)(row1 - row0) * mosaic_w * sizeof(uint16_t), (size_t)n_planes, ctx->stream));
    return MISTITCH_OK;
  }
  const int GX = ff_type == MISTITCH_FF_NONE ? kGxPlain : kGxFlat;
  // groups per row: plain mosaics count them from the aligned address at or below the row start (up to 7 elements
  // before it), chunk order from column 0
  const int groups_per_row = chunk_h != 0 ? out_w / 8 : (mosaic_w + 7 + 7) / 8;
  const dim3 grid((groups_per_row + 64 * GX * kWX - 1) / (64 * GX * kWX), (row_end - row0 + kBH - 1) / kBH, n_planes);
  MS_REQUIRE((int64_t)grid.x * grid.y * grid.z < (1ll << 30), MISTITCH_ERR_UNSUPPORTED, "mosaic too large for one launch");
  const int64_t n_blocks = (int64_t)grid.x * grid.y * grid.z;
  int rc;
  const size_t bins_bytes = (size_t)n_blocks * (kMaxCand + 1) * sizeof(int32_t);
  void* ws = static_ws;
  if (ws != nullptr) {
    MS_REQUIRE(bins_bytes + 16 <= kStaticScratchBytes, MISTITCH_ERR_UNSUPPORTED,
               "mistitch_fuse_static: %lld output blocks of %d x %d pixels need %zu bytes of tile lists, the static "
               "workspace holds %zu (about %lld blocks): fuse fewer rows per call, or call mistitch_fuse",
               (long long)n_blocks, kBH, 512 * GX * kWX, bins_bytes + 16, kStaticScratchBytes,
               (long long)((kStaticScratchBytes - 16) / ((kMaxCand + 1) * sizeof(int32_t))));
  } else {
    if ((rc = ctx->bins.reserve(bins_bytes + 16))) return rc;
    ws = ctx->bins.ptr;
  }
  // nothing in the workspace has to start at zero: k_fuse_bins writes every block's count and list and the
  // flatfield verdict before anything reads them
  const int32_t* bins = reinterpret_cast<const int32_t*>(ws);
  int32_t* ff_safe = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + bins_bytes);
  const int64_t n_bin_groups = (int64_t)grid.x * ((grid.y + kBinRows - 1) / kBinRows) * grid.z;
  MS_LAUNCH(ctx, "k_fuse_bins", k_fuse_bins, dim3((unsigned)((n_bin_groups + 3) / 4)), dim3(256), 0, n_tiles, tile_h, tile_w,
            tile_yx, tile_plane, reinterpret_cast<int32_t*>(ws), (int)grid.x, (int)grid.y, n_planes, mosaic_w,
            row0, row_end, 512 * GX * kWX, ff_type == MISTITCH_FF_F32 ? tile_ff : (const int32_t*)nullptr, ff_safe);
  if (ff_type == MISTITCH_FF_F32 && n_tiles > 0)
    MS_LAUNCH(ctx, "k_ff_check", k_ff_check, dim3(kFfChunks, kFfPlanes), dim3(256), 0,
              reinterpret_cast<const float*>(flatfields), 1, (int64_t)tile_h * tile_w, ff_safe);
  // Lanes of k_fuse without a plain copy still issue their 16-byte pixel load (and 32- or 64-byte flatfield load), at
  // the start of the tile / flatfield array.  An array of fewer than 8 values does not hold such a load: the idle
  // loads then go to the workspace instead (at least 148 bytes, 16-byte aligned), addressed relative to the arrays.
  int64_t idle_off = 0, idle_foff = 0;
  const size_t ff_elem = ff_type == MISTITCH_FF_F64 ? 8 : 4;
  if ((int64_t)(n_tiles - 1) * tile_pitch + (int64_t)tile_h * tile_w < 8) {
    MS_REQUIRE((reinterpret_cast<uintptr_t>(tiles) & 1u) == 0, MISTITCH_ERR_ARG, "tiles is not 2-byte aligned");
    idle_off = (int64_t)(reinterpret_cast<intptr_t>(ws) - reinterpret_cast<intptr_t>(tiles)) / 2;
  }
  if (ff_type != MISTITCH_FF_NONE && (int64_t)tile_h * tile_w < 8) {
    MS_REQUIRE(reinterpret_cast<uintptr_t>(flatfields) % ff_elem == 0, MISTITCH_ERR_ARG, "flatfields is not aligned to its element size");
    idle_foff = (int64_t)(reinterpret_cast<intptr_t>(ws) - reinterpret_cast<intptr_t>(flatfields)) / (int64_t)ff_elem;
  }
#define MS_FUSE(M, F)                                                                                          \
  launch_fuse<M, F, (F == MISTITCH_FF_NONE ? kGxPlain : kGxFlat)>(ctx, grid, tiles, n_tiles, tile_h, tile_w, tile_pitch, tile_yx, tile_plane, flatfields, tile_ff, \
                    bins, ff_safe, mosaic, mosaic_h, mosaic_w, row0, row_end, idle_off, idle_foff, cg)
  if (mode == MISTITCH_FUSE_OVERWRITE) {
    if (ff_type == MISTITCH_FF_NONE) MS_FUSE(MISTITCH_FUSE_OVERWRITE, MISTITCH_FF_NONE);
    else if (ff_type == MISTITCH_FF_F32) MS_FUSE(MISTITCH_FUSE_OVERWRITE, MISTITCH_FF_F32);
    else MS_FUSE(MISTITCH_FUSE_OVERWRITE, MISTITCH_FF_F64);
  } else {
    if (ff_type == MISTITCH_FF_NONE) MS_FUSE(MISTITCH_FUSE_FEATHER, MISTITCH_FF_NONE);
    else if (ff_type == MISTITCH_FF_F32) MS_FUSE(MISTITCH_FUSE_FEATHER, MISTITCH_FF_F32);
    else MS_FUSE(MISTITCH_FUSE_FEATHER, MISTITCH_FF_F64);
  }
#undef MS_FUSE
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_fuse(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                             int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                             const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* mosaic,
                             int n_planes, int mosaic_h, int mosaic_w, int row0, int row1, int mode) {
  return fuse_impl(ctx, tiles, n_tiles, tile_h, tile_w, tile_pitch, tile_yx, tile_plane, flatfields, ff_type, tile_ff,
                   mosaic, n_planes, mosaic_h, mosaic_w, row0, row1, mode, nullptr);
}

extern "C" int mistitch_fuse_chunked(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                                     int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                                     const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* chunks,
                                     int n_planes, int mosaic_h, int mosaic_w, int chunk_h, int chunk_w, int row0, int row1,
                                     int mode) {
  MS_REQUIRE(chunk_h > 0 && chunk_w > 0, MISTITCH_ERR_ARG, "bad chunk shape %d x %d", chunk_h, chunk_w);
  return fuse_impl(ctx, tiles, n_tiles, tile_h, tile_w, tile_pitch, tile_yx, tile_plane, flatfields, ff_type, tile_ff,
                   chunks, n_planes, mosaic_h, mosaic_w, row0, row1, mode, nullptr, chunk_h, chunk_w);
}

extern "C" int mistitch_fuse_static(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                                    int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                                    const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* mosaic,
                                    int n_planes, int mosaic_h, int mosaic_w, int row0, int row1, int mode) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(ctx->device >= 0 && ctx->device < kMaxDevices, MISTITCH_ERR_UNSUPPORTED, "device index %d beyond %d",
             ctx->device, kMaxDevices - 1);
  void* ws = __atomic_load_n(&g_scratch_of_device[ctx->device], __ATOMIC_ACQUIRE);
  MS_REQUIRE(ws != nullptr, MISTITCH_ERR_HIP, "the static workspace of device %d was never looked up", ctx->device);
  return fuse_impl(ctx, tiles, n_tiles, tile_h, tile_w, tile_pitch, tile_yx, tile_plane, flatfields, ff_type, tile_ff,
                   mosaic, n_planes, mosaic_h, mosaic_w, row0, row1, mode, ws);
}

extern "C" int mistitch_apply_flatfield(mistitch_ctx* ctx, const uint16_t* in, const void* flatfield, int ff_type,
                                        uint16_t* out, int n_tiles, int tile_h, int tile_w) {
  MS_REQUIRE(ctx && in && flatfield && out, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(n_tiles >= 0 && tile_h > 0 && tile_w > 0, MISTITCH_ERR_ARG, "bad geometry");
  MS_REQUIRE(ff_type == MISTITCH_FF_F32 || ff_type == MISTITCH_FF_F64, MISTITCH_ERR_ARG, "unknown flatfield type %d",
             ff_type);
  if (n_tiles == 0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  const int64_t npx = (int64_t)tile_h * tile_w;
  const int64_t gpt = npx / 8;
  const unsigned bpt = (unsigned)((gpt + 255) / 256);
  const int ntail = (int)(npx % 8) * n_tiles;
  int rc;
  if ((rc = ctx->flags.reserve(16))) return rc;
  int32_t* ff_safe = reinterpret_cast<int32_t*>(ctx->flags.ptr);
  if (ff_type == MISTITCH_FF_F32) {
    MS_HIP(hipMemsetAsync(ff_safe, 1, sizeof(int32_t), ctx->stream));
    MS_LAUNCH(ctx, "k_ff_check", k_ff_check, dim3(kFfChunks, 1), dim3(256), 0, reinterpret_cast<const float*>(flatfield),
              0, npx, ff_safe);
  }
  const unsigned tiles_per_launch = bpt ? std::max(1u, 0x7fffffffu / bpt) : 1u;
  for (unsigned t0 = 0; gpt && t0 < (unsigned)n_tiles; t0 += tiles_per_launch) {
    const unsigned nt = std::min(tiles_per_launch, (unsigned)n_tiles - t0);
    if (ff_type == MISTITCH_FF_F32)
      MS_LAUNCH(ctx, "k_flatfield", k_flatfield<MISTITCH_FF_F32>, dim3(nt * bpt), dim3(256), 0, in, flatfield, out, npx, bpt, t0, ff_safe, (unsigned)ctx->pixel_max);
    else
      MS_LAUNCH(ctx, "k_flatfield", k_flatfield<MISTITCH_FF_F64>, dim3(nt * bpt), dim3(256), 0, in, flatfield, out, npx, bpt, t0, ff_safe, (unsigned)ctx->pixel_max);
  }
  if (ntail) {
    if (ff_type == MISTITCH_FF_F32)
      MS_LAUNCH(ctx, "k_flatfield_tail", k_flatfield_tail<MISTITCH_FF_F32>, dim3((ntail + 63) / 64), dim3(64), 0, in, flatfield, out, npx, n_tiles, (unsigned)ctx->pixel_max);
    else
      MS_LAUNCH(ctx, "k_flatfield_tail", k_flatfield_tail<MISTITCH_FF_F64>, dim3((ntail + 63) / 64), dim3(64), 0, in, flatfield, out, npx, n_tiles, (unsigned)ctx->pixel_max);
  }
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

static int pyramid_impl(mistitch_ctx* ctx, const uint16_t* image, int n_planes, int h, int w, int n_levels,
                        double* const* levels, int chunk_h, int chunk_w, int out_chunk_h = 0, int out_chunk_w = 0) {
  MS_REQUIRE(ctx && image && levels, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(n_planes > 0 && h > 0 && w > 0, MISTITCH_ERR_ARG, "bad geometry");
  MS_REQUIRE(n_levels >= 1 && n_levels <= kPyrMaxLevels, MISTITCH_ERR_UNSUPPORTED, "n_levels must be in [1, %d]",
             kPyrMaxLevels);
  if (n_levels == 1) return MISTITCH_OK;  // level 0 is the image itself
  PyrLevels lv{};
  lv.och = out_chunk_h;
  lv.ocw = out_chunk_w;
  for (int l = 1; l < n_levels; ++l) {
    MS_REQUIRE(levels[l] != nullptr || (h >> l) == 0 || (w >> l) == 0, MISTITCH_ERR_ARG, "levels[%d] is NULL", l);
    lv.out[l] = levels[l];
  }
  MS_HIP(hipSetDevice(ctx->device));
  const dim3 grid((w + kPyrTileW - 1) / kPyrTileW, (h + kPyrTileH - 1) / kPyrTileH, n_planes);
  MS_REQUIRE(grid.y < 65536 && grid.z < 65536, MISTITCH_ERR_UNSUPPORTED, "image too large for one launch");
  ChunkGeom cg{0, 0, 0, 0};
  if (chunk_h != 0) {
    MS_REQUIRE(chunk_h > 0 && chunk_w > 0 && chunk_w % 8 == 0, MISTITCH_ERR_UNSUPPORTED,
               "chunks of %d x %d: the chunk width must be a multiple of 8 pixels", chunk_h, chunk_w);
    cg = ChunkGeom{chunk_h, chunk_w, (h + chunk_h - 1) / chunk_h, (w + chunk_w - 1) / chunk_w};
    MS_REQUIRE((int64_t)cg.ncx * chunk_h * chunk_w < (1ll << 31), MISTITCH_ERR_UNSUPPORTED, "a row of chunks of 2^31 elements or more");
  }
  MS_LAUNCH(ctx, "k_pyramid", k_pyramid, grid, dim3(256), 0, image, h, w, n_levels, lv, cg);
  if (out_chunk_h != 0)
    MS_LAUNCH(ctx, "k_pyramid_padding", k_pyramid_padding, dim3(1024, n_levels - 1, n_planes), dim3(256), 0, h, w, lv);
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_pyramid(mistitch_ctx* ctx, const uint16_t* image, int n_planes, int h, int w, int n_levels,
                                double* const* levels) {
  return pyramid_impl(ctx, image, n_planes, h, w, n_levels, levels, 0, 0);
}

extern "C" int mistitch_pyramid_chunked(mistitch_ctx* ctx, const uint16_t* chunks, int n_planes, int h, int w, int chunk_h,
                                        int chunk_w, int n_levels, double* const* levels) {
  MS_REQUIRE(chunk_h > 0 && chunk_w > 0, MISTITCH_ERR_ARG, "bad chunk shape %d x %d", chunk_h, chunk_w);
  return pyramid_impl(ctx, chunks, n_planes, h, w, n_levels, levels, chunk_h, chunk_w);
}

extern "C" int mistitch_pyramid_to_chunks(mistitch_ctx* ctx, const uint16_t* image, int n_planes, int h, int w,
                                          int chunk_h, int chunk_w, int image_is_chunked, int n_levels,
                                          double* const* levels) {
  MS_REQUIRE(chunk_h > 0 && chunk_w > 0, MISTITCH_ERR_ARG, "bad chunk shape %d x %d", chunk_h, chunk_w);
  MS_REQUIRE(n_planes > 0 && n_planes < 65536, MISTITCH_ERR_UNSUPPORTED, "%d planes", n_planes);
  return pyramid_impl(ctx, image, n_planes, h, w, n_levels, levels, image_is_chunked ? chunk_h : 0,
                      image_is_chunked ? chunk_w : 0, chunk_h, chunk_w);
}

// ---- chunk layout for the array writer ---------------------------------------------------------
// One workgroup per row of one chunk: chunk (plane, cy, cx), row r inside it.  The row's chunk_w
// elements are contiguous on both sides, so the copy moves 16-byte groups when both sides allow it
// (chunk widths in bytes multiples of 16 and an aligned output: the source rows may start anywhere, as the
// rows of a mosaic of odd width do, and are read with 16-byte loads at element alignment) and single elements
// otherwise; what sticks out of the image is written as zeros.
struct __attribute__((packed, aligned(2))) ChunkGroup2 {
  unsigned d[4];
};
struct __attribute__((packed, aligned(8))) ChunkGroup8 {
  unsigned d[4];
};
template <int ES, bool VEC>
__global__ __launch_bounds__(256) void k_chunk_layout(const unsigned char* __restrict__ image, int h, int w, int chunk_h,
                                                      int chunk_w, int ncy, int ncx, int64_t n_rows,
                                                      unsigned char* __restrict__ out) {
  const int64_t row_id = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;  // ((plane * ncy + cy) * ncx + cx) * chunk_h + r
  if (row_id >= n_rows) return;
  const int r = (int)(row_id % chunk_h);
  const int64_t chunk_id = row_id / chunk_h;
  const int cx = (int)(chunk_id % ncx);
  const int64_t pc = chunk_id / ncx;
  const int cy = (int)(pc % ncy);
  const int64_t plane = pc / ncy;
  const int y = cy * chunk_h + r, x0 = cx * chunk_w;
  const int valid = y < h ? min(chunk_w, w - x0) : 0;  // elements of this row that exist in the image
  const unsigned char* src = image + ((plane * h + y) * (int64_t)w + x0) * ES;
  unsigned char* dst = out + row_id * (int64_t)chunk_w * ES;
  if constexpr (VEC) {
    constexpr int kPer = 16 / ES;
    const int groups = chunk_w / kPer, full = valid / kPer;
    for (int g = threadIdx.x; g < groups; g += 256) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (g < full) {
        if constexpr (ES == 2) {
          const ChunkGroup2 q = *reinterpret_cast<const ChunkGroup2*>(src + (int64_t)g * 16);
          v = make_uint4(q.d[0], q.d[1], q.d[2], q.d[3]);
        } else {
          const ChunkGroup8 q = *reinterpret_cast<const ChunkGroup8*>(src + (int64_t)g * 16);
          v = make_uint4(q.d[0], q.d[1], q.d[2], q.d[3]);
        }
      } else if (g * kPer < valid) {  // the group the image's edge cuts
        unsigned char tmp[16] = {0};
        const int nb = (valid - g * kPer) * ES;
        for (int b = 0; b < nb; ++b) tmp[b] = src[(int64_t)g * 16 + b];
        v = *reinterpret_cast<const uint4*>(tmp);
      }
      __builtin_nontemporal_store(v.x, reinterpret_cast<unsigned*>(dst + (int64_t)g * 16));
      __builtin_nontemporal_store(v.y, reinterpret_cast<unsigned*>(dst + (int64_t)g * 16) + 1);
      __builtin_nontemporal_store(v.z, reinterpret_cast<unsigned*>(dst + (int64_t)g * 16) + 2);
      __builtin_nontemporal_store(v.w, reinterpret_cast<unsigned*>(dst + (int64_t)g * 16) + 3);
    }
  } else {
    for (int i = threadIdx.x; i < chunk_w; i += 256) {
      if constexpr (ES == 2) {
        reinterpret_cast<uint16_t*>(dst)[i] = i < valid ? reinterpret_cast<const uint16_t*>(src)[i] : (uint16_t)0;
      } else {
        reinterpret_cast<unsigned long long*>(dst)[i] = i < valid ? reinterpret_cast<const unsigned long long*>(src)[i] : 0ull;
      }
    }
  }
}

extern "C" int mistitch_chunk_layout(mistitch_ctx* ctx, const void* image, int elem_size, int n_planes, int h, int w,
                                     int chunk_h, int chunk_w, void* out) {
  MS_REQUIRE(ctx && image && out, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(elem_size == 2 || elem_size == 8, MISTITCH_ERR_UNSUPPORTED, "elements of %d bytes (uint16 and float64 planes only)",
             elem_size);
  MS_REQUIRE(n_planes > 0 && h > 0 && w > 0 && chunk_h > 0 && chunk_w > 0, MISTITCH_ERR_ARG, "bad geometry");
  const int ncy = (h + chunk_h - 1) / chunk_h, ncx = (w + chunk_w - 1) / chunk_w;
  const int64_t rows = (int64_t)n_planes * ncy * ncx * chunk_h;
  MS_REQUIRE(rows <= (65535ll << 20), MISTITCH_ERR_UNSUPPORTED, "%lld chunk rows exceed one launch", (long long)rows);
  MS_HIP(hipSetDevice(ctx->device));
  const bool vec = ((int64_t)chunk_w * elem_size) % 16 == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0 &&
                   (reinterpret_cast<uintptr_t>(image) % (uintptr_t)elem_size) == 0;
  const unsigned char* src = reinterpret_cast<const unsigned char*>(image);
  unsigned char* dst = reinterpret_cast<unsigned char*>(out);
  // one workgroup per chunk row, in a two-dimensional grid: a launch holds fewer than 2^32 threads along x
  constexpr int64_t kRowsX = 1 << 20;
  const dim3 grid((unsigned)std::min<int64_t>(rows, kRowsX), (unsigned)((rows + kRowsX - 1) / kRowsX));
  if (elem_size == 2) {
    if (vec) MS_LAUNCH(ctx, "k_chunk_layout", (k_chunk_layout<2, true>), grid, dim3(256), 0, src, h, w, chunk_h, chunk_w, ncy, ncx, rows, dst);
    else MS_LAUNCH(ctx, "k_chunk_layout", (k_chunk_layout<2, false>), grid, dim3(256), 0, src, h, w, chunk_h, chunk_w, ncy, ncx, rows, dst);
  } else {
    if (vec) MS_LAUNCH(ctx, "k_chunk_layout", (k_chunk_layout<8, true>), grid, dim3(256), 0, src, h, w, chunk_h, chunk_w, ncy, ncx, rows, dst);
    else MS_LAUNCH(ctx, "k_chunk_layout", (k_chunk_layout<8, false>), grid, dim3(256), 0, src, h, w, chunk_h, chunk_w, ncy, ncx, rows, dst);
  }
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

// ---- bitshuffle filter ---------------------------------------------------------------------------
// The reference's default chunk compressor is Blosc(cname='zstd', clevel=5, shuffle=BITSHUFFLE)
// (stitcher.py:536, :549).  Blosc cuts a chunk into blocks and, before the entropy coder sees a block, applies the
// bitshuffle filter to it: the block's whole groups of 8 elements are transposed as a bit matrix, so that
// out[(8 b + k) * (n / 8) + g] holds bit k of byte b of the elements 8 g .. 8 g + 7 (element 8 g + i in bit i), for
// n = the block's element count rounded down to a multiple of 8; the bytes after those n elements are copied.
// That filter is a streaming pass and is what this kernel does on the device; the entropy coder (zstd) stays with the
// caller's codec library.
// A workgroup takes 2048 elements of one block: lane t transposes the 8 x 8 bit matrices of its 8 elements (one per
// byte of the element) and leaves their rows in LDS, from where whole 16-byte pieces of the output rows are written.
constexpr int kShufElems = 2048;  // elements per workgroup
__device__ __forceinline__ unsigned long long transpose_bits_8x8(unsigned long long x) {
  unsigned long long t;
  t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;
  x = x ^ t ^ (t << 7);
  t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull;
  x = x ^ t ^ (t << 14);
  t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull;
  x = x ^ t ^ (t << 28);
  return x;
}

template <int ES>
__global__ __launch_bounds__(256) void k_bitshuffle(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                    int64_t n_bytes, int64_t block_bytes, int tiles_per_block, int vec_whole) {
  constexpr int kRows = 8 * ES;
  constexpr int kStride = 256 + 16;  // bytes per LDS row: 16-byte pieces stay aligned, rows fall into different banks
  __shared__ __align__(16) unsigned char rows[kRows * kStride];
  const int t = threadIdx.x;
  const int64_t blk = blockIdx.x / tiles_per_block;
  const int tile = blockIdx.x % tiles_per_block;
  const int64_t blk_off = blk * block_bytes;
  const int64_t bytes = min(block_bytes, n_bytes - blk_off);
  const bool vec = vec_whole && bytes == block_bytes;  // 16-byte accesses: whole blocks only (the last one may be short)
  const int64_t n_main = (bytes / ES) & ~(int64_t)7;  // elements that are transposed
  const int64_t row_bytes = n_main / 8;
  const unsigned char* src = in + blk_off;
  unsigned char* dst = out + blk_off;
  if (tile == 0) {  // what follows the whole groups of 8 elements is copied
    for (int64_t i = n_main * ES + t; i < bytes; i += 256) dst[i] = src[i];
  }
  const int64_t e0 = (int64_t)tile * kShufElems + 8 * t;  // first element of this lane
  if ((int64_t)tile * kShufElems >= n_main) return;        // uniform
  unsigned long long plane[ES];  // plane[b]: byte b of the lane's 8 elements, element i in byte i
#pragma unroll
  for (int b = 0; b < ES; ++b) plane[b] = 0ull;
  if (e0 < n_main) {
    const unsigned char* p = src + e0 * ES;
    if (vec) {  // 16-byte loads (the block starts are 16-byte aligned)
      unsigned w[2 * ES];
#pragma unroll
      for (int q = 0; q < ES / 2 + (ES == 1 ? 1 : 0); ++q) {
        if constexpr (ES >= 2) {
          const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * q);
          w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
        } else {
          const uint2 v = *reinterpret_cast<const uint2*>(p);
          w[0] = v.x, w[1] = v.y;
        }
      }
      // element i occupies bytes [i * ES, (i + 1) * ES) of w
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int b = 0; b < ES; ++b) {
          const int at = i * ES + b;
          plane[b] |= (unsigned long long)((w[at >> 2] >> (8 * (at & 3))) & 0xffu) << (8 * i);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int b = 0; b < ES; ++b) plane[b] |= (unsigned long long)p[i * ES + b] << (8 * i);
    }
  }
#pragma unroll
  for (int b = 0; b < ES; ++b) {
    const unsigned long long x = transpose_bits_8x8(plane[b]);
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[(8 * b + k) * kStride + t] = (unsigned char)(x >> (8 * k));
  }
  __syncthreads();
  // row r of the tile is bytes [tile * 256, +256) of output row r; 16 pieces of 16 bytes each
  const int64_t col0 = (int64_t)tile * (kShufElems / 8);
  for (int q = t; q < kRows * 16; q += 256) {
    const int r = q >> 4, seg = q & 15;
    const int64_t col = col0 + 16 * seg;
    if (col >= row_bytes) continue;
    unsigned char* o = dst + (int64_t)r * row_bytes + col;
    const unsigned char* l = rows + r * kStride + 16 * seg;
    if (vec && col + 16 <= row_bytes) {
      *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(l);
    } else {
      const int nb = (int)min((int64_t)16, row_bytes - col);
      for (int i = 0; i < nb; ++i) o[i] = l[i];
    }
  }
}

extern "C" int mistitch_bitshuffle(mistitch_ctx* ctx, const void* in, void* out, int64_t n_bytes, int elem_size,
                                   int64_t block_bytes) {
  MS_REQUIRE(ctx && in && out, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(in != out, MISTITCH_ERR_ARG, "the filter does not work in place");
  MS_REQUIRE(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8, MISTITCH_ERR_UNSUPPORTED,
             "elements of %d bytes", elem_size);
  MS_REQUIRE(n_bytes >= 0 && block_bytes > 0 && block_bytes % elem_size == 0, MISTITCH_ERR_ARG,
             "block size %lld is not a positive multiple of the element size", (long long)block_bytes);
  if (n_bytes == 0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  const int64_t n_blocks = (n_bytes + block_bytes - 1) / block_bytes;
  const int64_t block_elems = std::min(block_bytes, n_bytes) / elem_size;
  const int tiles = (int)std::max<int64_t>(1, (block_elems + kShufElems - 1) / kShufElems);
  MS_REQUIRE(n_blocks * tiles <= 0x7fffffffll, MISTITCH_ERR_UNSUPPORTED, "%lld workgroups exceed one launch",
             (long long)(n_blocks * tiles));
  // 16-byte accesses: every block starts on a 16-byte boundary and every output row of a whole block does too
  const int64_t main_full = (block_bytes / elem_size) & ~(int64_t)7;
  const int vec = (reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
                   block_bytes % 16 == 0 && (main_full / 8) % 16 == 0)
                      ? 1
                      : 0;
  const unsigned char* src = reinterpret_cast<const unsigned char*>(in);
  unsigned char* dst = reinterpret_cast<unsigned char*>(out);
  const dim3 grid((unsigned)(n_blocks * tiles));
#define MS_SHUF(E) MS_LAUNCH(ctx, "k_bitshuffle", k_bitshuffle<E>, grid, dim3(256), 0, src, dst, n_bytes, block_bytes, tiles, vec)
  if (elem_size == 1) MS_SHUF(1);
  else if (elem_size == 2) MS_SHUF(2);
  else if (elem_size == 4) MS_SHUF(4);
  else MS_SHUF(8);
#undef MS_SHUF
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_mip(mistitch_ctx* ctx, const uint16_t* in, int n, int64_t pitch, int64_t n_pixels,
                            uint16_t* out) {
  MS_REQUIRE(ctx && in && out, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(n >= 1, MISTITCH_ERR_ARG, "cannot project an empty stack");
  MS_REQUIRE(n_pixels >= 0 && pitch >= n_pixels, MISTITCH_ERR_ARG, "bad geometry");
  if (n_pixels == 0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_pixels / 8 + 255) / 256, 256 * 16));
  MS_LAUNCH(ctx, "k_mip", k_mip, dim3(blocks), dim3(256), 0, in, n, pitch, n_pixels, out);
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}
