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
struct __attribute__((packed, aligned(4))) F8 {
  float v[8];
};
struct __attribute__((packed, aligned(8))) D8 {
  double v[8];
};

struct TileRef {
  int ty, tx;  // mosaic position of the tile origin
  int t;       // tile index
  int ff;      // flatfield plane or -1
};

template <int FF>
__device__ __forceinline__ uint16_t flatfield_one(uint16_t v, const void* ffrow, int i) {
  if constexpr (FF == MISTITCH_FF_F32) {
    float q = __fdiv_rn((float)v, reinterpret_cast<const float*>(ffrow)[i]);
    q = fminf(fmaxf(q, 0.0f), 65535.0f);
    return (uint16_t)(unsigned int)q;
  } else if constexpr (FF == MISTITCH_FF_F64) {
    double q = (double)v / reinterpret_cast<const double*>(ffrow)[i];
    q = fmin(fmax(q, 0.0), 65535.0);
    return (uint16_t)(unsigned int)q;
  } else {
    return v;
  }
}

// Fetch the pixels of `m` (bit i = window pixel i) from tile row `src` (points
// at window pixel 0, which may lie outside the tile when bit 0 is clear).
template <int FF>
__device__ __forceinline__ void fetch_window(const uint16_t* src, const void* ffrow, unsigned m, uint16_t (&px)[8]) {
  if (m == 0xffu) {
    const Px8 v = *reinterpret_cast<const Px8*>(src);
    if constexpr (FF == MISTITCH_FF_F32) {
      const F8 f = *reinterpret_cast<const F8*>(ffrow);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float q = __fdiv_rn((float)v.at(i), f.v[i]);
        q = fminf(fmaxf(q, 0.0f), 65535.0f);
        px[i] = (uint16_t)(unsigned int)q;
      }
    } else if constexpr (FF == MISTITCH_FF_F64) {
      const D8 f = *reinterpret_cast<const D8*>(ffrow);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double q = (double)v.at(i) / f.v[i];
        q = fmin(fmax(q, 0.0), 65535.0);
        px[i] = (uint16_t)(unsigned int)q;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) px[i] = (uint16_t)v.at(i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (m & (1u << i)) px[i] = flatfield_one<FF>(src[i], ffrow, i);
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

// One wave per output block lists, in placement order, the tiles that touch it:
// bins[block][0] = how many (may exceed kMaxCand: the fuse kernel then scans every
// tile), bins[block][1..] = the first kMaxCand tile indices.
// grid = ceil(n_blocks / 4) workgroups of 4 waves; blocks are numbered as k_fuse's grid.
__global__ __launch_bounds__(256) void k_fuse_bins(int n_tiles, int H, int W, const int32_t* __restrict__ tile_yx,
                                                   const int32_t* __restrict__ tile_plane, int32_t* __restrict__ bins,
                                                   int gx, int gy, int n_planes, int Wm, int row0, int row1, int kBW) {
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= (int64_t)gx * gy * n_planes) return;
  const int bx = (int)(blk % gx), by = (int)((blk / gx) % gy), plane = (int)(blk / ((int64_t)gx * gy));
  const int x0 = max(bx * kBW - 7, 0), x1 = min(bx * kBW + kBW, Wm);
  const int y0 = row0 + by * kBH, y1 = min(y0 + kBH, row1);
  int32_t* bin = bins + blk * (kMaxCand + 1);
  int count = 0;
  for (int base = 0; base < n_tiles; base += 64) {
    const int t = base + lane;
    bool hit = false;
    if (t < n_tiles && tile_plane[t] == plane) {
      const int ty = tile_yx[2 * t], tx = tile_yx[2 * t + 1];
      hit = ty < y1 && ty + H > y0 && tx < x1 && tx + W > x0;
    }
    const unsigned long long mask = __ballot(hit);
    if (hit) {
      const int slot = count + __popcll(mask & ((1ull << lane) - 1ull));
      if (slot < kMaxCand) bin[1 + slot] = t;
    }
    count += __popcll(mask);
  }
  if (lane == 0) bin[0] = count;
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
                                               const int32_t* __restrict__ tile_ff, uint16_t (&px)[8]) {
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
        c = TileRef{tile_yx[2 * ci], tile_yx[2 * ci + 1], ci, tile_ff ? tile_ff[ci] : -1};
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
        fetch_window<FF>(src, ffrow, m, px);
      } else {
        fetch_window<MISTITCH_FF_NONE>(src, nullptr, m, px);
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
        c = TileRef{tile_yx[2 * ci], tile_yx[2 * ci + 1], ci, tile_ff ? tile_ff[ci] : -1};
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
        fetch_window<FF>(src, ffrow, m, v);
      } else {
        fetch_window<MISTITCH_FF_NONE>(src, nullptr, m, v);
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
          const float q = floorf(__fadd_rn(__fdiv_rn(num[i], den[i]), 0.5f));
          px[i] = (uint16_t)(unsigned int)fminf(q, 65535.0f);
        }
      }
    }
  }
}

// A group whose pixels come from several tiles, or only partly from one.  Feather
// groups that two to four tiles cover completely (the inside of an overlap band)
// are blended from loads issued together; everything else goes pixel by pixel.
template <int MODE, int FF>
__device__ __forceinline__ void slow_window(int y, int x, unsigned valid, const TileRef* list, int ncand, bool overflow,
                                            int plane, const uint16_t* __restrict__ tiles, int H, int W,
                                            int64_t tile_pitch, const int32_t* __restrict__ tile_yx,
                                            const int32_t* __restrict__ tile_plane,
                                            const void* __restrict__ flatfields,
                                            const int32_t* __restrict__ tile_ff, uint16_t (&px)[8]) {
  constexpr size_t ffsize = FF == MISTITCH_FF_F64 ? 8 : 4;
  if constexpr (MODE == MISTITCH_FUSE_FEATHER) {
    int touching = 0;
    bool all_full = valid == 0xffu && !overflow;
    unsigned packed = 0;  // up to four candidate indices, one per byte, in placement order
    if (all_full) {
      for (int ci = 0; ci < ncand; ++ci) {
        const TileRef c = list[ci];
        const int ly = y - c.ty, lx0 = x - c.tx;
        if ((unsigned)ly >= (unsigned)H) continue;
        const int lo = max(0, -lx0), hi = min(8, W - lx0);
        if (lo >= hi) continue;
        if (touching < 4) packed |= (unsigned)ci << (8 * touching);
        ++touching;
        all_full = all_full && lo == 0 && hi == 8 && (FF == MISTITCH_FF_NONE || c.ff >= 0);
      }
    }
    if (all_full && touching >= 2 && touching <= 4) {
      Px8 sv[4];
      F8 sf32[FF == MISTITCH_FF_F32 ? 4 : 1];
      D8 sf64[FF == MISTITCH_FF_F64 ? 4 : 1];
      int lys[4], lxs[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const TileRef c = list[(packed >> (8 * (q < touching ? q : 0))) & 0xffu];
        lys[q] = y - c.ty;
        lxs[q] = x - c.tx;
        const int64_t in_tile = (int64_t)lys[q] * W + lxs[q];
        sv[q] = *reinterpret_cast<const Px8*>(tiles + (int64_t)c.t * tile_pitch + in_tile);
        if constexpr (FF == MISTITCH_FF_F32)
          sf32[q] = *reinterpret_cast<const F8*>(reinterpret_cast<const char*>(flatfields) +
                                                 ((int64_t)c.ff * H * W + in_tile) * (int64_t)ffsize);
        if constexpr (FF == MISTITCH_FF_F64)
          sf64[q] = *reinterpret_cast<const D8*>(reinterpret_cast<const char*>(flatfields) +
                                                 ((int64_t)c.ff * H * W + in_tile) * (int64_t)ffsize);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        MS_PIN_PX8(sv[q]);
        if constexpr (FF == MISTITCH_FF_F32) MS_PIN_8(sf32[q]);
        if constexpr (FF == MISTITCH_FF_F64) MS_PIN_8(sf64[q]);
      }
      float num[8], den[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) num[i] = den[i] = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < touching) {
          const float wy = (float)min(lys[q] + 1, H - lys[q]);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float val;
            if constexpr (FF == MISTITCH_FF_F32) {
              float t = __fdiv_rn((float)sv[q].at(i), sf32[q].v[i]);
              t = fminf(fmaxf(t, 0.0f), 65535.0f);
              val = (float)(unsigned int)t;
            } else if constexpr (FF == MISTITCH_FF_F64) {
              double t = (double)sv[q].at(i) / sf64[q].v[i];
              t = fmin(fmax(t, 0.0), 65535.0);
              val = (float)(unsigned int)t;
            } else {
              val = (float)sv[q].at(i);
            }
            const int lx = lxs[q] + i;
            const float w = __fmul_rn(wy, (float)min(lx + 1, W - lx));
            num[i] = __fadd_rn(num[i], __fmul_rn(w, val));
            den[i] = __fadd_rn(den[i], w);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = floorf(__fadd_rn(__fdiv_rn(num[i], den[i]), 0.5f));
        px[i] = (uint16_t)(unsigned int)fminf(t, 65535.0f);
      }
      return;
    }
  }
  resolve_window<MODE, FF>(y, x, valid, list, ncand, overflow, plane, tiles, H, W, tile_pitch, tile_yx, tile_plane,
                           flatfields, tile_ff, px);
}

// 1-D grid of 8 * gx * ceil(gy * n_planes / 8) workgroups with
// gx = ceil(ceil((Wm + 7) / 8) / (64 GX)) and gy = ceil((row1 - row0) / kBH).
// GX = 16-byte groups per lane along a row: a wave moves GX KiB of one mosaic row per step.
template <int MODE, int FF, int GX>
__global__ __launch_bounds__(kFuseThreads) void k_fuse(const uint16_t* __restrict__ tiles, int n_tiles, int H,
                                                       int W, int64_t tile_pitch,
                                                       const int32_t* __restrict__ tile_yx,
                                                       const int32_t* __restrict__ tile_plane,
                                                       const void* __restrict__ flatfields,
                                                       const int32_t* __restrict__ tile_ff,
                                                       const int32_t* __restrict__ bins,
                                                       uint16_t* __restrict__ mosaic, int Hm, int Wm, int row0,
                                                       int row1, int gx, int gy, int n_planes) {
  __shared__ TileRef list[kMaxCand];
  __shared__ uint16_t queue[kBH * 64 * GX];  // (row in block << 8) | group in block row
  __shared__ unsigned queue_len;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) queue_len = 0;
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
  if (tid < total && tid < kMaxCand) {
    const int t = bin[1 + tid];
    list[tid] = TileRef{tile_yx[2 * t], tile_yx[2 * t + 1], t, tile_ff ? tile_ff[t] : -1};
  }
  __syncthreads();
  const bool overflow = total > kMaxCand;  // fall back to scanning every tile per group
  const int ncand = overflow ? n_tiles : total;
  uint16_t* mplane = mosaic + (int64_t)plane * Hm * Wm;
  const int w_idx = bx * (64 * GX) + lane;  // first aligned group of this lane within a row
  constexpr size_t ffsize = FF == MISTITCH_FF_F64 ? 8 : 4;

  // One mosaic row per wave and step.  A lane first classifies its GX groups
  // without touching pixel memory (0: uncovered, 1: plain copy from one tile,
  // 2: anything else), issues all copy loads back to back and stores.
  for (int y = y0 + wave; y < y1; y += kFuseThreads / 64) {
    uint16_t* rowp = mplane + (int64_t)y * Wm;
    const int a = (int)((reinterpret_cast<uintptr_t>(rowp) >> 1) & 7);
    Px8 v[GX];
    F8 f32[FF == MISTITCH_FF_F32 ? GX : 1];
    D8 f64[FF == MISTITCH_FF_F64 ? GX : 1];
    unsigned kinds = 0;  // two bits per group
#pragma unroll
    for (int r = 0; r < GX; ++r) {
      const int x = 8 * (w_idx + 64 * r) - a;
      unsigned vm = 0;
      if (x + 8 > 0 && x < Wm) vm = (((1u << min(8, Wm - x)) - 1u) & ~((1u << max(0, -x)) - 1u)) & 0xffu;
      int k = vm ? 0 : 3, sel = 0;  // 3: nothing to write
      if (vm && overflow) k = 2;
      if (vm && !overflow) {
        if constexpr (MODE == MISTITCH_FUSE_OVERWRITE) {
          for (int ci = ncand - 1; ci >= 0; --ci) {  // the latest tile that touches the group decides
            const TileRef c = list[ci];
            const int ly = y - c.ty, lx0 = x - c.tx;
            if ((unsigned)ly >= (unsigned)H) continue;
            const int lo = max(0, -lx0), hi = min(8, W - lx0);
            if (lo >= hi) continue;
            if (!((((1u << hi) - 1u) & ~((1u << lo) - 1u)) & vm)) continue;
            k = (lo == 0 && hi == 8 && vm == 0xffu && (FF == MISTITCH_FF_NONE || c.ff >= 0)) ? 1 : 2;
            sel = ci;
            break;
          }
        } else {
          int touching = 0;
          bool simple = vm == 0xffu;
          for (int ci = 0; ci < ncand; ++ci) {
            const TileRef c = list[ci];
            const int ly = y - c.ty, lx0 = x - c.tx;
            if ((unsigned)ly >= (unsigned)H) continue;
            const int lo = max(0, -lx0), hi = min(8, W - lx0);
            if (lo >= hi) continue;
            if (!((((1u << hi) - 1u) & ~((1u << lo) - 1u)) & vm)) continue;
            ++touching;
            sel = ci;
            simple = simple && lo == 0 && hi == 8 && (FF == MISTITCH_FF_NONE || c.ff >= 0);
          }
          if (touching == 1 && simple) k = 1;
          else if (touching >= 1) k = 2;
        }
      }
      kinds |= (unsigned)k << (2 * r);
      // copy loads never sit behind a branch: lanes without a plain copy read tile 0's first pixels
      int64_t off = 0, foff = 0;
      if (k == 1) {
        const TileRef c = list[sel];
        const int64_t in_tile = (int64_t)(y - c.ty) * W + (x - c.tx);
        off = (int64_t)c.t * tile_pitch + in_tile;
        foff = (int64_t)c.ff * H * W + in_tile;
      }
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
      if (k >= 2) continue;
      const int x = 8 * (w_idx + 64 * r) - a;
      uint16_t px[8];
      if (k == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (FF == MISTITCH_FF_F32) {
            float q = __fdiv_rn((float)v[r].at(i), f32[r].v[i]);
            q = fminf(fmaxf(q, 0.0f), 65535.0f);
            px[i] = (uint16_t)(unsigned int)q;
          } else if constexpr (FF == MISTITCH_FF_F64) {
            double q = (double)v[r].at(i) / f64[r].v[i];
            q = fmin(fmax(q, 0.0), 65535.0);
            px[i] = (uint16_t)(unsigned int)q;
          } else {
            px[i] = (uint16_t)v[r].at(i);
          }
        }
        store_window(rowp + x, px, 0xffu);
      } else {
        unsigned vm = (((1u << min(8, Wm - x)) - 1u) & ~((1u << max(0, -x)) - 1u)) & 0xffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) px[i] = 0;
        store_window(rowp + x, px, vm);
      }
    }
    // groups of the last kind are queued for the whole workgroup: they cluster in the
    // overlap bands, and handled in place they would stall every wave that owns a few
#pragma unroll
    for (int r = 0; r < GX; ++r) {
      if (((kinds >> (2 * r)) & 3u) == 2u) {
        const unsigned slot = atomicAdd(&queue_len, 1u);
        queue[slot] = (uint16_t)(((y - y0) << 8) | (lane + 64 * r));
      }
    }
  }
  __syncthreads();
  const unsigned n_slow = queue_len;
  for (unsigned i = tid; i < n_slow; i += kFuseThreads) {
    const unsigned e = queue[i];
    const int y = y0 + (int)(e >> 8);
    uint16_t* rowp = mplane + (int64_t)y * Wm;
    const int a = (int)((reinterpret_cast<uintptr_t>(rowp) >> 1) & 7);
    const int x = 8 * (bx * (64 * GX) + (int)(e & 0xffu)) - a;
    const unsigned vm = (((1u << min(8, Wm - x)) - 1u) & ~((1u << max(0, -x)) - 1u)) & 0xffu;
    uint16_t px[8];
    slow_window<MODE, FF>(y, x, vm, list, ncand, overflow, plane, tiles, H, W, tile_pitch, tile_yx, tile_plane,
                          flatfields, tile_ff, px);
    store_window(rowp + x, px, vm);
  }
}

// out = trunc(clip(in / ff)); one flatfield plane shared by n_tiles images.
template <int FF>
__global__ __launch_bounds__(256) void k_flatfield(const uint16_t* in, const void* __restrict__ ff, uint16_t* out,
                                                   int64_t n_px_per_tile, int64_t n_groups) {
  // n_groups = n_tiles * (n_px_per_tile / 8); the tail (n_px_per_tile % 8) is handled by k_flatfield_tail
  const int64_t gpt = n_px_per_tile / 8;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = g / gpt, q = g - t * gpt;
    const uint16_t* src = in + t * n_px_per_tile + q * 8;
    uint16_t px[8];
    const char* ffrow = reinterpret_cast<const char*>(ff) + q * 8 * (FF == MISTITCH_FF_F64 ? 8 : 4);
    fetch_window<FF>(src, ffrow, 0xffu, px);
    Px8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.d[i] = px[2 * i] | ((unsigned)px[2 * i + 1] << 16);
    *reinterpret_cast<Px8*>(out + t * n_px_per_tile + q * 8) = o;
  }
}

template <int FF>
__global__ void k_flatfield_tail(const uint16_t* in, const void* ff, uint16_t* out, int64_t n_px_per_tile,
                                 int n_tiles) {
  const int64_t tail0 = n_px_per_tile / 8 * 8;
  const int ntail = (int)(n_px_per_tile - tail0);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntail * n_tiles) return;
  const int t = i / ntail, e = i - t * ntail;
  const char* ffrow = reinterpret_cast<const char*>(ff) + tail0 * (FF == MISTITCH_FF_F64 ? 8 : 4);
  out[t * n_px_per_tile + tail0 + e] = flatfield_one<FF>(in[t * n_px_per_tile + tail0 + e], ffrow, e);
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

template <int MODE, int FF, int GX>
static void launch_fuse(mistitch_ctx* ctx, dim3 grid, const uint16_t* tiles, int n_tiles, int H, int W, int64_t pitch,
                        const int32_t* yx, const int32_t* plane, const void* ffs, const int32_t* tff,
                        const int32_t* bins, uint16_t* mosaic, int Hm, int Wm, int row0, int row1) {
  const int gx = (int)grid.x, gy = (int)grid.y, n_planes = (int)grid.z;
  const unsigned n_wg = 8u * gx * (unsigned)((gy * n_planes + 7) / 8);
  MS_LAUNCH(ctx, MODE == MISTITCH_FUSE_OVERWRITE ? "k_fuse_overwrite" : "k_fuse_feather", (k_fuse<MODE, FF, GX>),
            dim3(n_wg), dim3(kFuseThreads), 0, tiles, n_tiles, H, W, pitch, yx, plane, ffs, tff, bins, mosaic, Hm, Wm,
            row0, row1, gx, gy, n_planes);
}

}  // namespace mistitch

using namespace mistitch;

extern "C" int mistitch_fuse(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                             int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                             const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* mosaic,
                             int n_planes, int mosaic_h, int mosaic_w, int row0, int row1, int mode) {
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
  if (ff_type != MISTITCH_FF_NONE)
    MS_REQUIRE(flatfields && tile_ff, MISTITCH_ERR_ARG, "flatfield type given without flatfields / tile_ff");
  if (row1 == row0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  const int GX = ff_type == MISTITCH_FF_NONE ? kGxPlain : kGxFlat;
  const dim3 grid(((mosaic_w + 7 + 7) / 8 + 64 * GX - 1) / (64 * GX), (row1 - row0 + kBH - 1) / kBH, n_planes);
  MS_REQUIRE((int64_t)grid.x * grid.y * grid.z < (1ll << 30), MISTITCH_ERR_UNSUPPORTED, "mosaic too large for one launch");
  const int64_t n_blocks = (int64_t)grid.x * grid.y * grid.z;
  int rc;
  if ((rc = ctx->bins.reserve((size_t)n_blocks * (kMaxCand + 1) * sizeof(int32_t)))) return rc;
  const int32_t* bins = reinterpret_cast<const int32_t*>(ctx->bins.ptr);
  MS_LAUNCH(ctx, "k_fuse_bins", k_fuse_bins, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, n_tiles, tile_h, tile_w,
            tile_yx, tile_plane, reinterpret_cast<int32_t*>(ctx->bins.ptr), (int)grid.x, (int)grid.y, n_planes, mosaic_w,
            row0, row1, 512 * GX);
#define MS_FUSE(M, F)                                                                                          \
  launch_fuse<M, F, (F == MISTITCH_FF_NONE ? kGxPlain : kGxFlat)>(ctx, grid, tiles, n_tiles, tile_h, tile_w, tile_pitch, tile_yx, tile_plane, flatfields, tile_ff, \
                    bins, mosaic, mosaic_h, mosaic_w, row0, row1)
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

extern "C" int mistitch_apply_flatfield(mistitch_ctx* ctx, const uint16_t* in, const void* flatfield, int ff_type,
                                        uint16_t* out, int n_tiles, int tile_h, int tile_w) {
  MS_REQUIRE(ctx && in && flatfield && out, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(n_tiles >= 0 && tile_h > 0 && tile_w > 0, MISTITCH_ERR_ARG, "bad geometry");
  MS_REQUIRE(ff_type == MISTITCH_FF_F32 || ff_type == MISTITCH_FF_F64, MISTITCH_ERR_ARG, "unknown flatfield type %d",
             ff_type);
  if (n_tiles == 0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  const int64_t npx = (int64_t)tile_h * tile_w;
  const int64_t groups = (npx / 8) * n_tiles;
  const int blocks = (int)std::min<int64_t>((groups + 255) / 256, 256 * 16);
  const int ntail = (int)(npx % 8) * n_tiles;
  if (ff_type == MISTITCH_FF_F32) {
    if (groups) MS_LAUNCH(ctx, "k_flatfield", k_flatfield<MISTITCH_FF_F32>, dim3(blocks), dim3(256), 0, in, flatfield, out, npx, groups);
    if (ntail) MS_LAUNCH(ctx, "k_flatfield_tail", k_flatfield_tail<MISTITCH_FF_F32>, dim3((ntail + 63) / 64), dim3(64), 0, in, flatfield, out, npx, n_tiles);
  } else {
    if (groups) MS_LAUNCH(ctx, "k_flatfield", k_flatfield<MISTITCH_FF_F64>, dim3(blocks), dim3(256), 0, in, flatfield, out, npx, groups);
    if (ntail) MS_LAUNCH(ctx, "k_flatfield_tail", k_flatfield_tail<MISTITCH_FF_F64>, dim3((ntail + 63) / 64), dim3(64), 0, in, flatfield, out, npx, n_tiles);
  }
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
