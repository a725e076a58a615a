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
constexpr int kBW = 512;      // block width in pixels: one wave covers a row of it
constexpr int kBH = 32;       // block height in pixels
constexpr int kMaxCand = 32;  // tiles remembered per block before the slow path

struct __attribute__((packed, aligned(2))) Px8 {
  uint16_t v[8];
};
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
        float q = __fdiv_rn((float)v.v[i], f.v[i]);
        q = fminf(fmaxf(q, 0.0f), 65535.0f);
        px[i] = (uint16_t)(unsigned int)q;
      }
    } else if constexpr (FF == MISTITCH_FF_F64) {
      const D8 f = *reinterpret_cast<const D8*>(ffrow);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double q = (double)v.v[i] / f.v[i];
        q = fmin(fmax(q, 0.0), 65535.0);
        px[i] = (uint16_t)(unsigned int)q;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) px[i] = v.v[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (m & (1u << i)) px[i] = flatfield_one<FF>(src[i], ffrow, i);
  }
}

// One wave per output block lists, in placement order, the tiles that touch it:
// bins[block][0] = how many (may exceed kMaxCand: the fuse kernel then scans every
// tile), bins[block][1..] = the first kMaxCand tile indices.
// grid = ceil(n_blocks / 4) workgroups of 4 waves; blocks are numbered as k_fuse's grid.
__global__ __launch_bounds__(256) void k_fuse_bins(int n_tiles, int H, int W, const int32_t* __restrict__ tile_yx,
                                                   const int32_t* __restrict__ tile_plane, int32_t* __restrict__ bins,
                                                   int gx, int gy, int n_planes, int Wm, int row0, int row1) {
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= (int64_t)gx * gy * n_planes) return;
  const int bx = (int)(blk % gx), by = (int)((blk / gx) % gy), plane = (int)(blk / ((int64_t)gx * gy));
  const int x0 = bx * kBW, x1 = min(x0 + kBW, Wm);
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

// grid (ceil(Wm / kBW), ceil((row1 - row0) / kBH), n_planes)
template <int MODE, int FF>
__global__ __launch_bounds__(kFuseThreads) void k_fuse(const uint16_t* __restrict__ tiles, int n_tiles, int H,
                                                       int W, int64_t tile_pitch,
                                                       const int32_t* __restrict__ tile_yx,
                                                       const int32_t* __restrict__ tile_plane,
                                                       const void* __restrict__ flatfields,
                                                       const int32_t* __restrict__ tile_ff,
                                                       const int32_t* __restrict__ bins,
                                                       uint16_t* __restrict__ mosaic, int Hm, int Wm, int row0,
                                                       int row1) {
  __shared__ TileRef list[kMaxCand];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * kBW, x1 = min(x0 + kBW, Wm);
  const int y0 = row0 + blockIdx.y * kBH, y1 = min(y0 + kBH, row1);
  // the tiles touching this block, in placement order, were listed by k_fuse_bins
  const int32_t* bin = bins + ((int64_t)(plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (kMaxCand + 1);
  const int total = bin[0];
  if (tid < total && tid < kMaxCand) {
    const int t = bin[1 + tid];
    list[tid] = TileRef{tile_yx[2 * t], tile_yx[2 * t + 1], t, tile_ff ? tile_ff[t] : -1};
  }
  __syncthreads();
  const int ncand_all = total;
  const bool overflow = ncand_all > kMaxCand;  // fall back to scanning every tile per window
  const int ncand = overflow ? n_tiles : ncand_all;
  const size_t ffsize = FF == MISTITCH_FF_F64 ? 8 : 4;

  uint16_t* mplane = mosaic + (int64_t)plane * Hm * Wm;
  for (int y = y0 + wave; y < y1; y += kFuseThreads / 64) {
    const int x = x0 + 8 * lane;
    if (x >= x1) continue;
    const int nvalid = min(8, Wm - x);
    const unsigned valid = (1u << nvalid) - 1u;
    uint16_t px[8];
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
          const char* ffrow = reinterpret_cast<const char*>(flatfields) +
                              (((int64_t)c.ff * H + ly) * W + lx0) * (int64_t)ffsize;
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
          const char* ffrow = reinterpret_cast<const char*>(flatfields) +
                              (((int64_t)c.ff * H + ly) * W + lx0) * (int64_t)ffsize;
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
    uint16_t* dst = mplane + (int64_t)y * Wm + x;
    if (nvalid == 8) {
      Px8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.v[i] = px[i];
      *reinterpret_cast<Px8*>(dst) = o;
    } else {
      for (int i = 0; i < nvalid; ++i) dst[i] = px[i];
    }
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
    for (int i = 0; i < 8; ++i) o.v[i] = px[i];
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
      for (int i = 0; i < 8; ++i) best.v[i] = v.v[i] > best.v[i] ? v.v[i] : best.v[i];
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

template <int MODE, int FF>
static void launch_fuse(mistitch_ctx* ctx, dim3 grid, const uint16_t* tiles, int n_tiles, int H, int W, int64_t pitch,
                        const int32_t* yx, const int32_t* plane, const void* ffs, const int32_t* tff,
                        const int32_t* bins, uint16_t* mosaic, int Hm, int Wm, int row0, int row1) {
  MS_LAUNCH(ctx, MODE == MISTITCH_FUSE_OVERWRITE ? "k_fuse_overwrite" : "k_fuse_feather", (k_fuse<MODE, FF>), grid,
            dim3(kFuseThreads), 0, tiles, n_tiles, H, W, pitch, yx, plane, ffs, tff, bins, mosaic, Hm, Wm, row0, row1);
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
  MS_REQUIRE(n_planes <= 65535, MISTITCH_ERR_UNSUPPORTED, "more than 65535 planes");
  if (row1 == row0) return MISTITCH_OK;
  MS_HIP(hipSetDevice(ctx->device));
  const dim3 grid((mosaic_w + kBW - 1) / kBW, (row1 - row0 + kBH - 1) / kBH, n_planes);
  MS_REQUIRE(grid.y <= 65535, MISTITCH_ERR_UNSUPPORTED, "row band too tall for one launch");
  const int64_t n_blocks = (int64_t)grid.x * grid.y * grid.z;
  int rc;
  if ((rc = ctx->bins.reserve((size_t)n_blocks * (kMaxCand + 1) * sizeof(int32_t)))) return rc;
  const int32_t* bins = reinterpret_cast<const int32_t*>(ctx->bins.ptr);
  MS_LAUNCH(ctx, "k_fuse_bins", k_fuse_bins, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, n_tiles, tile_h, tile_w,
            tile_yx, tile_plane, reinterpret_cast<int32_t*>(ctx->bins.ptr), (int)grid.x, (int)grid.y, n_planes, mosaic_w,
            row0, row1);
#define MS_FUSE(M, F)                                                                                          \
  launch_fuse<M, F>(ctx, grid, tiles, n_tiles, tile_h, tile_w, tile_pitch, tile_yx, tile_plane, flatfields, tile_ff, \
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
