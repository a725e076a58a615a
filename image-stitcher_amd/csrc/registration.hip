// Pairwise registration kernels and their C entry points.
//
// Canonical strip layout.  Every strip is processed as [S][L] with the long
// axis L contiguous: a "top" strip (w rows of the tile) is read in place with
// S = w, L = tile width; a "left" strip (w columns of the tile) is transposed
// once into a staging buffer with S = w, L = tile height.  The 2-D DFT, the
// phase-correlation peak set and every NCC moment are invariant under that
// transposition; only the reported (row, col) are swapped back.
//
// Pipeline per pass of pairs (all kernels skip pairs whose status is set):
//   k_setup_pairs      descriptors from (pair_idx, pair_dir)
//   k_edge_stats       exact variance of the 16-strided sample -> featureless
//   k_transpose_strips left strips -> staging [S][L]
//   k_fft_long_fwd     z = a + i b per line, length-L FFT, digit-reversed out
//   k_fft_short_cps    length-S FFT down columns, split into the two spectra,
//                      normalised cross-power spectrum, inverse length-S FFT;
//                      only the Hermitian half of the columns is written back
//   k_fft_long_inv     two lines per complex inverse FFT, scale, top-k per block
//   k_peaks_candidates merge block top-k, emit the 32 signed candidates
//   k_ncc_accumulate   exact integer moments of every candidate overlap
//   k_finalize         NCC from the moments, first strict maximum wins
#include <algorithm>
#include <cmath>
#include <limits>
#include <set>

#include "common.hpp"
#include "fft.hpp"
#include "fft_reg.hpp"

namespace mistitch {

#ifdef MS_STAMPS
// Diagnostic build only: phase time stamps of the FFT kernels (never compiled into the shipped library).
__device__ unsigned long long g_stamps[3][4096][8];
#define MS_STAMP(kern, slot)                                                                    \
  do {                                                                                          \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_stamps[kern][blockIdx.x][slot] = wall_clock64(); \
  } while (0)
#else
#define MS_STAMP(kern, slot) \
  do {                       \
  } while (0)
#endif

constexpr int kThreads = 256;
constexpr int kMaxK = 16;       // peaks a block / a pair can report
constexpr int kNumCand = 32;    // 2 peaks x 16 signed interpretations: the discrete search of the default path
constexpr int kNumCandWide = 160;  // 10 peaks x 16: the candidate set of the reference's consensus stage (MISTITCH_REG_PEAKS10)
constexpr int kNccChunks = 8;   // workgroups cooperating on one candidate (2, 4 and 16 are all slower on 2048 x 256 strips)
// A pair has n_cand + 1 candidate slots (n_cand = kNumCand or kNumCandWide, a kernel argument): the last slot
// holds the subpixel estimate's rounded shift (flag 1 when unused).
constexpr int kUpsMax = 160;    // largest upsampled region side (ceil(1.5 * upsample factor))
constexpr int kUpsNodes = 32;   // Chebyshev nodes per axis at which the subpixel stage evaluates the correlation (two 16-row tiles)
constexpr int kUpsRows = 16;    // rows of the upsampled grid per workgroup of the interpolation kernel (one matrix tile)

struct PairDesc {
  const uint16_t* a;  // canonical strip of the first tile
  const uint16_t* b;  // canonical strip of the second tile
  int64_t pitch;      // elements between canonical lines
  int S, L;           // canonical extents
  int transposed;     // 1: canonical (s, l) = image (col, row)
  int img_h, img_w;   // strip extents in image orientation
  int off_y, off_x;   // strip origin inside the first tile
};

struct Peak {
  double val;
  int key;  // flat index in image orientation; -1 = none
  int loc;  // block-local location (selection bookkeeping)
};

// Subpixel refinement state of a pair (all along the canonical axes).
struct UpsState {
  int shift_s, shift_l;  // coarse shift: signed position of the largest |PCM| value
  int valid;             // 0: no usable coarse peak
  int pad;
  double sub_s, sub_l;   // refined shift
};

struct Cand {
  int y, x;    // image orientation, strip relative
  int flag;    // 0 evaluate, 1 empty overlap (skipped), 2 fewer than 25 pixels, 3 duplicate
  int dup_of;  // index of the earlier identical candidate when flag == 3
};

// Rows of the intermediate spectrum are kept 128 bytes longer than L: with a
// power-of-two pitch every row of a column block maps to the same few cache
// sets and memory channels, and the column passes run several times slower.
template <typename T>
__host__ __device__ __forceinline__ int spectrum_pitch(int L) {
  return L + 128 / (int)(2 * sizeof(T));
}

// Between the first long-axis pass and the column pass the spectrum is kept in
// SLABS: the kSlab = 8 adjacent storage positions of every line are gathered into
// one contiguous [S][8] piece (plus 128 bytes of padding), element (s, pos) at
// (pos / 8) * slab_pitch + s * 8 + pos % 8.  The column pass then reads whole
// slabs contiguously; the scattered accesses are all on the WRITING side, where
// the memory system can batch them.
// Largest radix of the long-axis plans: complex128 lines run radix-8 passes (one more pass,
// but 90 instead of 154 VGPRs: five waves per SIMD instead of three), complex64 ones radix 16.
#ifndef MS_LONG_RADIX_F64
#define MS_LONG_RADIX_F64 8
#endif
template <typename T>
constexpr int kLongMaxRadix = sizeof(T) == 8 ? MS_LONG_RADIX_F64 : 16;

// Column pass: complex128 blocks run 512 lanes with radix-8 plans (under 128 VGPRs, four waves
// per SIMD), complex64 ones 256 lanes with radix 16 (their LDS footprint already allows four
// workgroups per CU).
template <typename T>
constexpr int kShortMaxRadix = sizeof(T) == 8 ? 8 : 16;
template <typename T>
constexpr int kShortThreads = 512;

constexpr int kSlab = 8;
// (an odd S gets one more row: the subpixel stage wants an even count, with an empty first row then)
__host__ __device__ __forceinline__ int64_t slab_pitch(int S) { return (int64_t)(S + (S & 1)) * kSlab + kSlab; }
__host__ __device__ __forceinline__ int64_t slab_pair_elems(int S, int L) { return slab_pitch(S) * (L / kSlab); }

// (bitwise operators on purpose, here and in first_better: short-circuit ones become a chain of
// divergent branches around every comparison, several times the cost of the comparisons)
__device__ __forceinline__ bool peak_better(double v, int k, double bv, int bk) {
  return (v > bv) | ((v == bv) & (k > bk));
}

// ---- wave-wide reductions ---------------------------------------------------
// Data-parallel primitives (DPP) and lane reads only.  The cross-lane shuffles of __shfl_xor go
// through the LDS pipeline; with several values per wave and thousands of short workgroups that
// pipeline, not memory, set the pace of the NCC kernel (2.3 ms with shuffles, 1.5 ms without).
// Four DPP steps leave every lane of a 16-lane row with the row's result; the four rows are
// combined through lane reads.
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // the other quad of each 8 lanes
constexpr int kDppMirror = 0x140;      // the other half of each 16 lanes

// Sum of a 64-bit value over the 64 lanes of a wave, the same in every lane.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xf, 0xf, true);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  v += dpp_u64<kDppXor1>(v);
  v += dpp_u64<kDppXor2>(v);
  v += dpp_u64<kDppHalfMirror>(v);
  v += dpp_u64<kDppMirror>(v);
  unsigned long long total = 0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, 16 * row);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), 16 * row);
    total += ((unsigned long long)hi << 32) | lo;
  }
  return total;
}

// The same for a 32-bit value whose wave total fits 32 bits: one DPP add per step.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += dpp_u32<kDppXor1>(v);
  v += dpp_u32<kDppXor2>(v);
  v += dpp_u32<kDppHalfMirror>(v);
  v += dpp_u32<kDppMirror>(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}


template <int CTRL>
__device__ __forceinline__ Peak dpp_peak(const Peak& p) {
  Peak o;
  o.val = __longlong_as_double((long long)dpp_u64<CTRL>((unsigned long long)__double_as_longlong(p.val)));
  o.key = (int)__builtin_amdgcn_update_dpp(0u, (unsigned)p.key, CTRL, 0xf, 0xf, true);
  o.loc = (int)__builtin_amdgcn_update_dpp(0u, (unsigned)p.loc, CTRL, 0xf, 0xf, true);
  return o;
}
__device__ __forceinline__ Peak lane_peak(const Peak& p, int lane) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(p.val);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)bits, lane), hi = __builtin_amdgcn_readlane((unsigned)(bits >> 32), lane);
  Peak o;
  o.val = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  o.key = (int)__builtin_amdgcn_readlane((unsigned)p.key, lane);
  o.loc = (int)__builtin_amdgcn_readlane((unsigned)p.loc, lane);
  return o;
}
// Wave-wide best peak under a total order `better(v, k, bv, bk)`, the same in every lane.
template <class Better>
__device__ __forceinline__ Peak wave_best_peak(Peak mine, Better better) {
  auto take = [&](const Peak& o) {
    if (better(o.val, o.key, mine.val, mine.key)) mine = o;
  };
  take(dpp_peak<kDppXor1>(mine));
  take(dpp_peak<kDppXor2>(mine));
  take(dpp_peak<kDppHalfMirror>(mine));
  take(dpp_peak<kDppMirror>(mine));
  Peak best = lane_peak(mine, 0);
#pragma unroll
  for (int row = 1; row < 4; ++row) {
    const Peak o = lane_peak(mine, 16 * row);
    if (better(o.val, o.key, best.val, best.key)) best = o;
  }
  return best;
}

// Among equal values the SMALLEST key wins (the first occurrence in row-major order, as
// numpy.argmax reports).
__device__ __forceinline__ bool first_better(double v, int k, double bv, int bk) {
  return (v > bv) | ((v == bv) & (k >= 0) & ((bk < 0) | (k < bk)));
}

// Block-wide argmax of (val, key) carrying loc; result valid in every thread.
__device__ __forceinline__ Peak block_argmax(Peak mine, Peak* red /*[kThreads/64]*/) {
  mine = wave_best_peak(mine, peak_better);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = mine;
  __syncthreads();
  Peak best = red[0];
#pragma unroll
  for (int w = 1; w < kThreads / 64; ++w) {
    Peak o = red[w];
    if (peak_better(o.val, o.key, best.val, best.key)) best = o;
  }
  return best;
}

// Block-wide argmax where the smallest key wins among equal values; result valid in every thread.
template <int NW = kThreads / 64>
__device__ __forceinline__ Peak block_argmax_first(Peak mine, Peak* red /*[NW]*/) {
  mine = wave_best_peak(mine, first_better);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = mine;
  __syncthreads();
  Peak best = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    Peak o = red[w];
    if (first_better(o.val, o.key, best.val, best.key)) best = o;
  }
  return best;
}

// ---------------------------------------------------------------------------
__global__ void k_setup_pairs(const uint16_t* tiles, int64_t tile_pitch, int H, int W, int w_left,
                              int w_top, const int32_t* pair_idx, const uint8_t* pair_dir, int n_pairs,
                              int n_tiles, uint16_t* staging, int pair_chunk, PairDesc* desc,
                              int8_t* status, int64_t slot_elems, int stage_top) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  int i1 = pair_idx[2 * p], i2 = pair_idx[2 * p + 1];
  PairDesc d;
  int st = MISTITCH_PAIR_OK;
  if (i1 < 0 || i2 < 0 || i1 >= n_tiles || i2 >= n_tiles) {
    st = MISTITCH_PAIR_FEATURELESS;  // unusable pair: reported like a failed one
    i1 = i2 = 0;
  }
  const uint16_t* t1 = tiles + (int64_t)i1 * tile_pitch;
  const uint16_t* t2 = tiles + (int64_t)i2 * tile_pitch;
  if (pair_dir[p] == 0) {  // left: last w columns of t1 against first w columns of t2
    int slot = p % pair_chunk;
    uint16_t* st_a = staging + (int64_t)slot * slot_elems;
    d.a = st_a;
    d.b = st_a + (int64_t)w_left * H;
    d.pitch = H;
    d.S = w_left;
    d.L = H;
    d.transposed = 1;
    d.img_h = H;
    d.img_w = w_left;
    d.off_y = 0;
    d.off_x = W - w_left;
  } else {  // top: last w rows of t1 against first w rows of t2
    d.a = t1 + (int64_t)(H - w_top) * W;
    d.b = t2;
    if (stage_top) {  // flatfield-corrected copies in the staging buffer
      uint16_t* st_a = staging + (int64_t)(p % pair_chunk) * slot_elems;
      d.a = st_a;
      d.b = st_a + (int64_t)w_top * W;
    }
    d.pitch = W;
    d.S = w_top;
    d.L = W;
    d.transposed = 0;
    d.img_h = w_top;
    d.img_w = W;
    d.off_y = H - w_top;
    d.off_x = 0;
  }
  // the reference's peak ranking refuses a correlation matrix with a side below 2
  // (_validate_pcm_input, _translation_computation.py:444-477, called by multi_peak_max) and the pair fails like a featureless one
  if (d.S < 2 || d.L < 2) st = MISTITCH_PAIR_FEATURELESS;
  desc[p] = d;
  status[p] = (int8_t)st;
}

// Exact population variance of strip[::16, ::16] from integer moments.
__device__ __forceinline__ double sampled_variance(const uint16_t* base, int64_t pitch, int h, int w,
                                                   unsigned long long* red) {
  const int nr = (h + 15) / 16, nc = (w + 15) / 16;
  unsigned long long s = 0, q = 0;
  for (int e = threadIdx.x; e < nr * nc; e += blockDim.x) {
    int r = e / nc, c = e - r * nc;
    unsigned long long v = base[(int64_t)(16 * r) * pitch + 16 * c];
    s += v;
    q += v * v;
  }
  s = wave_sum_u64(s);
  q = wave_sum_u64(q);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[2 * wave] = s;
    red[2 * wave + 1] = q;
  }
  __syncthreads();
  s = q = 0;
  for (int i = 0; i < (int)blockDim.x / 64; ++i) {
    s += red[2 * i];
    q += red[2 * i + 1];
  }
  const double n = (double)(nr * nc);
  const double mean = (double)s / n;
  return (double)q / n - mean * mean;
}

// grid (2, n_pairs): which strip, pair.
__global__ void k_edge_stats(const uint16_t* tiles, int64_t tile_pitch, int H, int W, int w_left,
                             int w_top, const int32_t* pair_idx, const uint8_t* pair_dir,
                             int8_t* status, double threshold) {
  __shared__ unsigned long long red[2 * kThreads / 64];
  const int p = blockIdx.y, which = blockIdx.x;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const uint16_t* t = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch;
  const uint16_t* base;
  int h, w;
  if (pair_dir[p] == 0) {
    h = H;
    w = w_left;
    base = which == 0 ? t + (W - w_left) : t;
  } else {
    h = w_top;
    w = W;
    base = which == 0 ? t + (int64_t)(H - w_top) * W : t;
  }
  double var = sampled_variance(base, W, h, w, red);
  if (threadIdx.x == 0 && var < threshold) status[p] = MISTITCH_PAIR_FEATURELESS;
}

__global__ void k_variance_single(const uint16_t* strip, int64_t pitch, int h, int w, double* out) {
  __shared__ unsigned long long red[2 * kThreads / 64];
  double var = sampled_variance(strip, pitch, h, w, red);
  if (threadIdx.x == 0) out[0] = var;
}

// Left strips into canonical [S][L]: staging[s][l] = tile[l][x0 + s].
// grid (ceil(L/32), ceil(S/32), 2 * pairs_in_pass), block (32, 8).
__global__ void k_transpose_strips(const uint16_t* tiles, int64_t tile_pitch, int H, int W, int w_left,
                                   const int32_t* pair_idx, const PairDesc* desc, const int8_t* status,
                                   int pair0) {
  __shared__ uint16_t tile[32][33];
  const int p = pair0 + (blockIdx.z >> 1), which = blockIdx.z & 1;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (!d.transposed) return;
  const uint16_t* src = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch + (which == 0 ? W - w_left : 0);
  uint16_t* dst = const_cast<uint16_t*>(which == 0 ? d.a : d.b);
  const int l0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    int l = l0 + r, s = s0 + threadIdx.x;
    if (l < H && s < w_left) tile[r][threadIdx.x] = src[(int64_t)l * W + s];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    int s = s0 + r, l = l0 + threadIdx.x;
    if (l < H && s < w_left) dst[(int64_t)s * H + l] = tile[threadIdx.x][r];
  }
}

// Streamed once: the intermediate spectra are written by one pass and read by the next, a whole pass of pairs later.
// MS_REG_NT (bit mask) marks those accesses non-temporal: 1 = stores of the forward long pass, 2 = loads and
// 4 = stores of the column pass, 8 = loads of the inverse long pass, 16 = strip stores and 32 = tile loads of the
// staging kernel, 64 = strip loads of the forward long pass, 128 = slab loads of the subpixel stage's short pass,
// 256 = the column pass's stores of that stage's copy.  Measured on 760 pairs of 2048 x 256 strips, same box:
// 5.40 ms per step with none, 5.28 with 1 (the column pass that follows gains, 1.87 -> 1.72 ms), 5.19 with 1 + 2 + 8;
// 4 costs the column pass more (1.97 ms) than it gives the inverse pass.  With a flatfield (every strip staged): 5.52 ms
// with 11, 5.45 with 16 or 32 or both added, 5.50 with 64 added.  With the subpixel stage: 7.52 ms with 59, 7.35 with 256 added
// (the column pass 2.24 -> 2.13 ms), 128 changes nothing.
#ifndef MS_REG_NT
#define MS_REG_NT 315
#endif
typedef double nt_d2 __attribute__((ext_vector_type(2)));
template <int BIT>
__device__ __forceinline__ double2 stream_load(const double2* p) {
  if constexpr ((MS_REG_NT & BIT) != 0) {
    const nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p));
    return make_double2(v.x, v.y);
  } else {
    return *p;
  }
}
template <int BIT>
__device__ __forceinline__ void stream_store(double2* p, double2 v) {
  if constexpr ((MS_REG_NT & BIT) != 0) {
    __builtin_nontemporal_store(nt_d2{v.x, v.y}, reinterpret_cast<nt_d2*>(p));
  } else {
    *p = v;
  }
}

typedef float nt_f2 __attribute__((ext_vector_type(2)));
template <int BIT>
__device__ __forceinline__ float2 stream_load(const float2* p) {
  if constexpr ((MS_REG_NT & BIT) != 0) {
    const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(p));
    return make_float2(v.x, v.y);
  } else {
    return *p;
  }
}
template <int BIT>
__device__ __forceinline__ void stream_store(float2* p, float2 v) {
  if constexpr ((MS_REG_NT & BIT) != 0) {
    __builtin_nontemporal_store(nt_f2{v.x, v.y}, reinterpret_cast<nt_f2*>(p));
  } else {
    *p = v;
  }
}
typedef unsigned nt_u4r __attribute__((ext_vector_type(4)));
template <int BIT>
__device__ __forceinline__ uint4 stream_load_u4(const uint4* p) {
  if constexpr ((MS_REG_NT & BIT) != 0) {
    const nt_u4r v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4r*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}
template <int BIT>
__device__ __forceinline__ void stream_store_u4(uint4* p, uint4 v) {
  if constexpr ((MS_REG_NT & BIT) != 0) {
    __builtin_nontemporal_store(nt_u4r{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u4r*>(p));
  } else {
    *p = v;
  }
}

// Same for strips whose sides are multiples of 64 and whose rows are 16-byte
// aligned: 64 x 64 tiles, 16 bytes per lane in and out.
// EDGE: the short side (the edge width) is arbitrary: the strip of the first tile then starts at an unaligned
// column (16-byte loads at 2-byte alignment), and the last 64 x 64 tile is cut by the strip's edge.
// grid (L / 64, ceil(S / 64), 2 * pairs_in_pass), block 256.
struct __attribute__((packed, aligned(2))) StripGroup {
  unsigned d[4];
};
template <bool EDGE>
__global__ __launch_bounds__(256) void k_transpose_strips64(const uint16_t* tiles, int64_t tile_pitch, int H, int W,
                                                            int w_left, const int32_t* pair_idx, const PairDesc* desc,
                                                            const int8_t* status, int pair0) {
  __shared__ uint16_t tile[64][64 + 2];
  const int p = pair0 + (blockIdx.z >> 1), which = blockIdx.z & 1;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (!d.transposed) return;
  const uint16_t* src = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch + (which == 0 ? W - w_left : 0);
  uint16_t* dst = const_cast<uint16_t*>(which == 0 ? d.a : d.b);
  const int l0 = blockIdx.x * 64, s0 = blockIdx.y * 64;
  const int t = threadIdx.x, sub = t & 7, row = t >> 3;  // 8 lanes cover one 128-byte row segment
  uint4 v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint16_t* at = src + (int64_t)(l0 + row + 32 * i) * W + s0 + 8 * sub;
    if constexpr (EDGE) {
      const int left = w_left - (s0 + 8 * sub);  // columns of this group inside the strip
      if (left >= 8) {
        const StripGroup g = *reinterpret_cast<const StripGroup*>(at);
        v[i] = make_uint4(g.d[0], g.d[1], g.d[2], g.d[3]);
      } else {  // the group the strip's edge cuts (or one beyond it): nothing is read outside the strip
        unsigned px[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) px[q] = q < left ? at[q] : 0u;
        v[i] = make_uint4(px[0] | (px[1] << 16), px[2] | (px[3] << 16), px[4] | (px[5] << 16), px[6] | (px[7] << 16));
      }
    } else {
      v[i] = stream_load_u4<32>(reinterpret_cast<const uint4*>(at));
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned w4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    unsigned* dstw = reinterpret_cast<unsigned*>(&tile[row + 32 * i][8 * sub]);  // rows are 132 bytes: 4-byte aligned
#pragma unroll
    for (int q = 0; q < 4; ++q) dstw[q] = w4[q];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = row + 32 * i;  // output line
    unsigned w4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      w4[q] = (unsigned)tile[8 * sub + 2 * q][s] | ((unsigned)tile[8 * sub + 2 * q + 1][s] << 16);
    uint4 o;
    o.x = w4[0];
    o.y = w4[1];
    o.z = w4[2];
    o.w = w4[3];
    if (!EDGE || s0 + s < w_left) stream_store_u4<16>(reinterpret_cast<uint4*>(dst + (int64_t)(s0 + s) * H + l0 + 8 * sub), o);
  }
}

// trunc(clip(v / ff)) for one pixel (tile_registration.py:2142-2181); FF as in mistitch.h.
template <int FF>
__device__ __forceinline__ unsigned ff_corrected(unsigned v, const void* ff, int64_t idx, unsigned vmax) {
  if constexpr (FF == MISTITCH_FF_F32) {
    const float q = __fdiv_rn((float)v, reinterpret_cast<const float*>(ff)[idx]);
    return (unsigned)fminf(fmaxf(q, 0.0f), (float)vmax);
  } else {
    const double q = (double)v / reinterpret_cast<const double*>(ff)[idx];
    return (unsigned)fmin(fmax(q, 0.0), (double)vmax);
  }
}

// trunc(clip(v / f)) with the flatfield value in hand.
template <int FF, typename F>
__device__ __forceinline__ unsigned ff_corrected_by(unsigned v, F f, unsigned vmax) {
  if constexpr (FF == MISTITCH_FF_F32) {
    const float q = __fdiv_rn((float)v, (float)f);
    return (unsigned)fminf(fmaxf(q, 0.0f), (float)vmax);
  } else {
    const double q = (double)v / (double)f;
    return (unsigned)fmin(fmax(q, 0.0), (double)vmax);
  }
}

// Strip extraction with the flatfield applied on the way: both strips of every pair go to
// the staging buffer in canonical [S][L] layout (left strips transposed), corrected as
// load_batch_images corrects whole tiles before the reference cuts its edges
// (tile_registration.py:2184-2279).  Only a quarter of the pixels of a tile lie in strips.
// Long sides multiples of 64, 16-byte aligned rows.  EDGE: the short side (the edge width) is arbitrary;
// the left strips then start at an unaligned column and are read pixel by pixel.
// Every pair of a direction cuts the same pixels of its tiles, so the flatfield values of a 64 x 64 piece of a strip
// are the same for all of them: a workgroup keeps its 16 values per lane in registers and takes the pairs of a
// group of kStagePairs in turn (the flatfield is then read once per group, not once per pair).
// grid (max L / 64, ceil(max S / 64), 4 * ceil(pairs_in_pass / kStagePairs)): z = 4 group + 2 direction + strip.
#ifndef MS_STAGE_PAIRS
#define MS_STAGE_PAIRS 16
#endif
constexpr int kStagePairs = MS_STAGE_PAIRS;
template <int FF, bool EDGE>
__global__ __launch_bounds__(256) void k_stage_strips64(const uint16_t* tiles, int64_t tile_pitch, int H, int W,
                                                        int w_left, int w_top, const int32_t* pair_idx,
                                                        const PairDesc* desc, const int8_t* status, int pair0,
                                                        int n_in_pass, const void* __restrict__ ff, unsigned vmax) {
  using FT = typename std::conditional<FF == MISTITCH_FF_F32, float, double>::type;
  __shared__ uint16_t tile[64][64 + 2];
  const int which = blockIdx.z & 1, dir = (blockIdx.z >> 1) & 1;  // dir 1: left pairs (strips transposed)
  const int p_begin = pair0 + (int)(blockIdx.z >> 2) * kStagePairs, p_end = min(pair0 + n_in_pass, p_begin + kStagePairs);
  const int S = dir ? w_left : w_top, L = dir ? H : W;
  const int l0 = blockIdx.x * 64, s0 = blockIdx.y * 64;
  if (l0 >= L || s0 >= S) return;
  const FT* ffv = reinterpret_cast<const FT*>(ff);
  const int t = threadIdx.x, sub = t & 7, row = t >> 3;  // 8 lanes cover one 128-byte row segment
  if (dir) {
    // image rows l0 .. l0+63, image columns x0 + s0 .. +63
    const int x0 = (which == 0 ? W - w_left : 0) + s0;
    FT fr[16];
    if constexpr (EDGE) {
      const int c = t & 63;
#pragma unroll
      for (int i = 0; i < 16; ++i) fr[i] = ffv[(int64_t)(l0 + (t >> 6) + 4 * i) * W + x0 + (s0 + c < S ? c : 0)];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) fr[8 * i + q] = ffv[(int64_t)(l0 + row + 32 * i) * W + x0 + 8 * sub + q];
    }
    // the next pair of this direction in the group (p_end: none); the same for every lane
    auto next_pair = [&](int p) {
      while (p < p_end && (status[p] != MISTITCH_PAIR_OK || desc[p].transposed != dir)) ++p;
      return p;
    };
    uint4 cur[2] = {}, nxt[2] = {};  // (aligned path) the pair's pixels, requested one pair ahead
    auto fetch = [&](int p, uint4 (&v)[2]) {
      const uint16_t* img = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch;
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = stream_load_u4<32>(reinterpret_cast<const uint4*>(img + (int64_t)(l0 + row + 32 * i) * W + x0 + 8 * sub));
    };
    int p = next_pair(p_begin);
    if (!EDGE && p < p_end) fetch(p, cur);
    for (; p < p_end;) {
      const int pn = next_pair(p + 1);
      if (!EDGE && pn < p_end) fetch(pn, nxt);
      const PairDesc d = desc[p];
      const uint16_t* img = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch;
      uint16_t* dst = const_cast<uint16_t*>(which == 0 ? d.a : d.b);
      if constexpr (EDGE) {
        const int c = t & 63;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (t >> 6) + 4 * i;
          const int64_t at = (int64_t)(l0 + r) * W + x0 + c;
          tile[r][c] = s0 + c < S ? (uint16_t)ff_corrected_by<FF>(img[at], fr[i], vmax) : (uint16_t)0;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 v = cur[i];
          const unsigned w4[4] = {v.x, v.y, v.z, v.w};
          unsigned* dstw = reinterpret_cast<unsigned*>(&tile[row + 32 * i][8 * sub]);  // rows are 132 bytes: 4-byte aligned
#pragma unroll
          for (int q = 0; q < 4; ++q)
            dstw[q] = ff_corrected_by<FF>(w4[q] & 0xffffu, fr[8 * i + 2 * q], vmax) |
                      (ff_corrected_by<FF>(w4[q] >> 16, fr[8 * i + 2 * q + 1], vmax) << 16);
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int sline = row + 32 * i;  // output line
        if (EDGE && s0 + sline >= S) continue;
        uint4 o;
        unsigned w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          w4[q] = (unsigned)tile[8 * sub + 2 * q][sline] | ((unsigned)tile[8 * sub + 2 * q + 1][sline] << 16);
        o.x = w4[0];
        o.y = w4[1];
        o.z = w4[2];
        o.w = w4[3];
        stream_store_u4<16>(reinterpret_cast<uint4*>(dst + (int64_t)(s0 + sline) * H + l0 + 8 * sub), o);
      }
      __syncthreads();  // the piece has been read out: the next pair may overwrite it
      cur[0] = nxt[0];
      cur[1] = nxt[1];
      p = pn;
    }
  } else {
    const int y0 = (which == 0 ? H - w_top : 0) + s0;
    FT fr[16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sline = min(row + 32 * i, S - 1 - s0);  // (EDGE: lines past the strip are not stored)
#pragma unroll
      for (int q = 0; q < 8; ++q) fr[8 * i + q] = ffv[(int64_t)(y0 + sline) * W + l0 + 8 * sub + q];
    }
    auto next_pair = [&](int p) {
      while (p < p_end && (status[p] != MISTITCH_PAIR_OK || desc[p].transposed != dir)) ++p;
      return p;
    };
    uint4 cur[2] = {}, nxt[2] = {};  // the pair's pixels, requested one pair ahead (lines past the strip: the last line's)
    auto fetch = [&](int p, uint4 (&v)[2]) {
      const uint16_t* img = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        v[i] = stream_load_u4<32>(reinterpret_cast<const uint4*>(img + (int64_t)(y0 + min(row + 32 * i, S - 1 - s0)) * W + l0 + 8 * sub));
    };
    int p = next_pair(p_begin);
    if (p < p_end) fetch(p, cur);
    for (; p < p_end;) {
      const int pn = next_pair(p + 1);
      if (pn < p_end) fetch(pn, nxt);
      const PairDesc d = desc[p];
      uint16_t* dst = const_cast<uint16_t*>(which == 0 ? d.a : d.b);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int sline = row + 32 * i;
        if (EDGE && s0 + sline >= S) continue;
        const uint4 v = cur[i];
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
        uint4 o;
        unsigned r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          r4[q] = ff_corrected_by<FF>(w4[q] & 0xffffu, fr[8 * i + 2 * q], vmax) |
                  (ff_corrected_by<FF>(w4[q] >> 16, fr[8 * i + 2 * q + 1], vmax) << 16);
        o.x = r4[0];
        o.y = r4[1];
        o.z = r4[2];
        o.w = r4[3];
        stream_store_u4<16>(reinterpret_cast<uint4*>(dst + (int64_t)(s0 + sline) * W + l0 + 8 * sub), o);
      }
      cur[0] = nxt[0];
      cur[1] = nxt[1];
      p = pn;
    }
  }
}

// The same for any strip size: one pixel per thread step.  grid (blocks, 2 * pairs_in_pass).
template <int FF>
__global__ __launch_bounds__(256) void k_stage_strips_any(const uint16_t* tiles, int64_t tile_pitch, int H, int W,
                                                          int w_left, int w_top, const int32_t* pair_idx,
                                                          const PairDesc* desc, const int8_t* status, int pair0,
                                                          const void* __restrict__ ff, unsigned vmax) {
  const int p = pair0 + (blockIdx.y >> 1), which = blockIdx.y & 1;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  const uint16_t* img = tiles + (int64_t)pair_idx[2 * p + which] * tile_pitch;
  uint16_t* dst = const_cast<uint16_t*>(which == 0 ? d.a : d.b);
  const int64_t n = (int64_t)d.S * d.L;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int sline = (int)(e / d.L), l = (int)(e - (int64_t)sline * d.L);
    int y, x;
    if (d.transposed) {
      y = l;
      x = (which == 0 ? W - w_left : 0) + sline;
    } else {
      y = (which == 0 ? H - w_top : 0) + sline;
      x = l;
    }
    const int64_t at = (int64_t)y * W + x;
    dst[e] = (uint16_t)ff_corrected<FF>(img[at], ff, at, vmax);
  }
}

// The featureless test on staged (canonical) strips.  grid (2, pairs_in_pass).
__global__ void k_edge_stats_staged(const PairDesc* desc, int8_t* status, int pair0, double threshold) {
  __shared__ unsigned long long red[2 * kThreads / 64];
  const int p = pair0 + blockIdx.y, which = blockIdx.x;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  // every 16th sample in both directions: the same set in canonical and in image orientation
  const double var = sampled_variance(which == 0 ? d.a : d.b, d.pitch, d.S, d.L, red);
  if (threadIdx.x == 0 && var < threshold) status[p] = MISTITCH_PAIR_FEATURELESS;
}

// Generic transposing copy used by the single-strip entry points (tall strips).
__global__ void k_transpose_plain(const uint16_t* src, int64_t pitch, int h, int w, uint16_t* dst) {
  __shared__ uint16_t tile[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    int y = r0 + r, x = c0 + threadIdx.x;
    if (y < h && x < w) tile[r][threadIdx.x] = src[(int64_t)y * pitch + x];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    int x = c0 + r, y = r0 + threadIdx.x;
    if (y < h && x < w) dst[(int64_t)x * h + y] = tile[threadIdx.x][r];
  }
}

// ---------------------------------------------------------------------------
// Forward FFT along the long axis.  grid (S / NF, pairs_in_pass).
// WIN: every pixel is multiplied by a separable window, win_s[line] * win_l[column] (float64 tables), on its
// way into the transform (MISTITCH_REG_WINDOW_HANN; the reference applies none).
template <typename T, bool VEC, bool WIN = false>
__global__ __launch_bounds__(kThreads) void k_fft_long_fwd(const PairDesc* desc, const int8_t* status,
                                                           int pair0, int S, int L, int want_transposed,
                                                           FftPlanDev planL, int NF, cx<T>* slabs,
                                                           const double* __restrict__ win_s = nullptr,
                                                           const double* __restrict__ win_l = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cx<T>* buf = reinterpret_cast<cx<T>*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  const int tid = threadIdx.x;
  LineAddr<T> addr{LineAddr<T>::padded(L) + 1};
  MS_STAMP(0, 0);
  const Twiddles<T> tw = load_twiddles<T>(buf + NF * addr.fstride, planL, tid, kThreads);
  const int s0 = blockIdx.x * NF;
  const int lL = 31 - __builtin_clz(L), lNF = 31 - __builtin_clz(NF);  // powers of two on this path
  if constexpr (VEC) {
    constexpr int U = 2;  // 16-byte loads of both strips in flight before the first LDS write
    for (int e0 = tid * 8; e0 < NF * L; e0 += kThreads * 8 * U) {
      uint4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // out-of-range lanes re-read the last group instead of branching: a branch
        // around a load makes the compiler drain the memory pipe at every join
        const int e = min(e0 + u * kThreads * 8, NF * L - 8);
        const int f = e >> lL, i = e & (L - 1);
        va[u] = *reinterpret_cast<const uint4*>(d.a + (int64_t)(s0 + f) * d.pitch + i);
        vb[u] = *reinterpret_cast<const uint4*>(d.b + (int64_t)(s0 + f) * d.pitch + i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        MS_PIN(va[u].x); MS_PIN(va[u].y); MS_PIN(va[u].z); MS_PIN(va[u].w);
        MS_PIN(vb[u].x); MS_PIN(vb[u].y); MS_PIN(vb[u].z); MS_PIN(vb[u].w);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads * 8;
        if (e < NF * L) {
          const int f = e >> lL, i = e & (L - 1);
          const unsigned wa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, wb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
          if constexpr (WIN) {
            const double ws = win_s[s0 + f];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double w0 = ws * win_l[i + 2 * q], w1 = ws * win_l[i + 2 * q + 1];
              buf[addr(f, i + 2 * q)] = mk<T>((T)((double)(wa[q] & 0xffffu) * w0), (T)((double)(wb[q] & 0xffffu) * w0));
              buf[addr(f, i + 2 * q + 1)] = mk<T>((T)((double)(wa[q] >> 16) * w1), (T)((double)(wb[q] >> 16) * w1));
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              buf[addr(f, i + 2 * q)] = mk<T>((T)(wa[q] & 0xffffu), (T)(wb[q] & 0xffffu));
              buf[addr(f, i + 2 * q + 1)] = mk<T>((T)(wa[q] >> 16), (T)(wb[q] >> 16));
            }
          }
        }
      }
    }
  } else {
    for (int e = tid; e < NF * L; e += kThreads) {
      const int f = e >> lL, i = e & (L - 1);
      const T va = (T)d.a[(int64_t)(s0 + f) * d.pitch + i], vb = (T)d.b[(int64_t)(s0 + f) * d.pitch + i];
      if constexpr (WIN) {
        const double w = win_s[s0 + f] * win_l[i];
        buf[addr(f, i)] = mk<T>((T)((double)va * w), (T)((double)vb * w));
      } else {
        buf[addr(f, i)] = mk<T>(va, vb);
      }
    }
  }
  __syncthreads();
  MS_STAMP(0, 1);
  fft_forward<T, kLongMaxRadix<T>>(buf, addr, NF, planL, tw, tid, kThreads);
  MS_STAMP(0, 2);
  // element (s0 + f, pos) -> slab pos / 8; the NF lines of this block are adjacent in every slab
  cx<T>* out = slabs + (int64_t)blockIdx.y * slab_pair_elems(S, L) + (int64_t)s0 * kSlab;
  const int64_t sp = slab_pitch(S);
  for (int e = tid; e < NF * L; e += kThreads) {
    const int c8 = e % kSlab, f = (e / kSlab) & (NF - 1), slab = (e / kSlab) >> lNF;
    stream_store<1>(&out[slab * sp + f * kSlab + c8], buf[addr(f, slab * kSlab + c8)]);
  }
  MS_STAMP(0, 3);
}

// The same pass for complex128 lines of N = 512 .. 4096 points on the register-resident engine (fft_reg.hpp): a
// lane takes its 8 pixels of either strip straight from memory (a wave reads 128 contiguous bytes per
// instruction), the twiddle bases come from the plan's table of roots with the same round trip, and the only
// LDS traffic is the exchanges between the radix passes and the transposition in front of the slab stores.
// grid (S, pairs_in_pass), N / 8 lanes, one line per workgroup.
template <int N, bool WIN>
__global__ __launch_bounds__(N / 8) void k_fft_long_fwd_reg(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                            int want_transposed, const double2* __restrict__ roots,
                                                            double2* slabs, const double* __restrict__ win_s,
                                                            const double* __restrict__ win_l) {
  using F = RegFft<N>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  const int8_t pair_status = status[p];
  const PairDesc d = desc[p];
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int s = blockIdx.x;
  // every load that depends on nothing but the descriptor goes out together
  const typename F::Tw tw = F::load_twiddles(roots, t);
  if (pair_status != MISTITCH_PAIR_OK || d.S != S || d.L != N || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  const uint16_t* ra = d.a + (int64_t)s * d.pitch + t;
  const uint16_t* rb = d.b + (int64_t)s * d.pitch + t;
  unsigned pa[8], pb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr ((MS_REG_NT & 64) != 0) {
      pa[j] = __builtin_nontemporal_load(ra + F::NL * j);
      pb[j] = __builtin_nontemporal_load(rb + F::NL * j);
    } else {
      pa[j] = ra[F::NL * j];
      pb[j] = rb[F::NL * j];
    }
  }
  double2 v[8];
  if constexpr (WIN) {
    const double ws = win_s[s];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double wgt = ws * win_l[t + F::NL * j];
      v[j] = mk<double>((double)pa[j] * wgt, (double)pb[j] * wgt);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = mk<double>((double)pa[j], (double)pb[j]);
  }
  F::forward(v, buf, tw, t);
  // the wave's own 512 storage positions, 64 consecutive ones per trip: 8 lanes fill one 128-byte slab piece
  double2* out = slabs + (int64_t)blockIdx.y * slab_pair_elems(S, N) + (int64_t)s * kSlab;
  const int64_t sp = slab_pitch(S);
  const double2* from = buf + F::slot(w * 512 + l);
  double2* to = out + (int64_t)(w * 64 + (l >> 3)) * sp + (l & 7);
#pragma unroll
  for (int i = 0; i < 8; ++i) stream_store<1>(&to[(int64_t)(8 * i) * sp], from[72 * i]);
}

// The spectra of two strips of INTEGER pixels are integers at the frequencies that are their own negatives (0 or half
// the length along both axes): |value| <= 65535 S L < 2^53, and a float64 transform is off by far less than 0.5 there,
// so rounding recovers them exactly - as numpy's transforms deliver them for lengths of radix 2, 3 and 4.  It matters
// when such a sum is exactly zero (an alternating sum of pixels that happens to cancel): the normalised cross-power
// value is then 0, where the rounding residue of a transform would be normalised to magnitude 1 with an arbitrary sign
// (1 / (S L) of every PCM value).  Not applied to windowed (non-integer) strips.
__device__ __forceinline__ void exact_real_bin(double& a, double& b) {
  a = rint(a);
  b = rint(b);
}
__device__ __forceinline__ void exact_real_bin(float&, float&) {}  // complex64: the sums exceed 24 bits

template <typename T>
__device__ __forceinline__ T cabs_safe(T x, T y);
template <>
__device__ __forceinline__ double cabs_safe<double>(double x, double y) {
  return sqrt(x * x + y * y);  // |x|,|y| <= 2^67: no overflow in double
}
template <>
__device__ __forceinline__ float cabs_safe<float>(float x, float y) {
  return hypotf(x, y);  // products of two 2^33 sums overflow float when squared
}

// A unit of the column pass: the storage positions [a0, a0 + na) whose results
// are computed and written back, plus the positions [b0, b0 + nb) that hold their
// negated frequencies (nb = 0 when the first range is closed under negation).
struct CpsUnit {
  int a0, na, b0, nb;
};

// Short-axis FFT, cross-power spectrum, short-axis inverse FFT.
// grid (n_units, pairs_in_pass), NT lanes.  The unit's [S][2C] block fills the LDS budget of
// half a CU whatever NT is, so a complex128 pass runs 512 lanes a workgroup: four waves per
// SIMD instead of two hide the LDS round trip of each radix pass (2.6 instead of 3.1 ms on
// the 2048 x 256 strips).
//
// MIXED: S is not a power of two but a product of radices up to 13 (complex128 only).  The passes
// are the division-based ones, the twiddles one table of all S roots; rows of the subpixel stage's
// copy start at row Sr / 2 - S / 2 of Sr = S + (S & 1) rows, the first of which stays empty for odd S.
template <typename T, int NT, int MAXR, bool MIXED = false>
__global__ __launch_bounds__(NT) void k_fft_short_cps(const PairDesc* desc, const int8_t* status, int pair0, int S, int L,
                                                      int want_transposed, FftPlanDev planS, FftPlanDev planL,
                                                      const CpsUnit* __restrict__ units, int rs, const cx<T>* slabs,
                                                      cx<T>* spectrum, T eps, cx<T>* keep_r,
                                                      const uint16_t* __restrict__ pos2stored, int n_stored, int exact) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cx<T>* buf = reinterpret_cast<cx<T>*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  // every lookup that does not depend on another goes out before the first test: one memory round trip
  // instead of three in front of the column loads
  const int8_t pair_status = status[p];
  const int d_S = desc[p].S, d_L = desc[p].L, d_transposed = desc[p].transposed;
  const CpsUnit un = units[blockIdx.x];
  if (pair_status != MISTITCH_PAIR_OK) return;
  if (d_S != S || d_L != L || (want_transposed >= 0 && d_transposed != want_transposed)) return;
  const int na = un.na, ncol = un.na + un.nb;
  const int tid = threadIdx.x;
  const int Lp = spectrum_pitch<T>(n_stored);
  cx<T>* Z = spectrum + (int64_t)blockIdx.y * S * Lp;
  const int64_t sp = slab_pitch(S);
  const cx<T>* Zs = slabs + (int64_t)blockIdx.y * slab_pair_elems(S, L);
  MS_STAMP(1, 0);
  static_assert(!MIXED || sizeof(T) == 8, "the mixed-radix passes are complex128");
  Twiddles<T> tw{nullptr, nullptr};
  cx<T>* twN = buf + S * rs;  // MIXED: all S roots
  if constexpr (MIXED) {
    const cx<T>* src = reinterpret_cast<const cx<T>*>(planS.tw);
    for (int q = tid; q < S; q += NT) twN[q] = src[q];
  } else {
    tw = load_twiddles<T>(buf + S * rs, planS, tid, NT);
  }
  // the short-axis index tables go to LDS once instead of being looked up in memory per cell
  uint16_t* mposS = reinterpret_cast<uint16_t*>(buf + S * rs + (MIXED ? S : Twiddles<T>::entries(S)));
  uint16_t* rowS = mposS + S;  // position -> row of the kept spectrum (ascending signed frequency)
  for (int q = tid; q < S; q += NT) {
    mposS[q] = planS.mpos[q];
    const int f = planS.pos2freq[q];
    if constexpr (MIXED)
      rowS[q] = (uint16_t)((f < (S + 1) / 2 ? f : f - S) + (S + (S & 1)) / 2);
    else
      rowS[q] = (uint16_t)((f + S / 2) & (S - 1));
  }
  const int lncol = 31 - __builtin_clz(ncol), lna = 31 - __builtin_clz(na);  // 8 or 16 columns: shifts, not divisions
  const int n_elem = S * ncol;
  // looked up here, with the column loads behind them, not where they are used: column c of a
  // lane's cells is the same on every trip (na is a power of two no larger than NT)
  const int mp = planL.mpos[un.a0 + (tid & (na - 1))];
  const int col0 = pos2stored[un.a0];
  ColAddr<T> addr{rs};
  // complex128 power-of-two columns: the first radix pass reads the slabs itself and the last one writes the
  // spectrum itself (2.57 -> 2.38 ms on the 2048 x 256 strips; the pass is bound by LDS traffic).  complex64
  // columns lose by it (1.45 -> 2.37 ms: 8-byte accesses, sixteen to a radix-16 butterfly) and keep the
  // separate copy loops.
  constexpr bool kFuseFirst = sizeof(T) == 8;
  constexpr bool kFuseLast = kFuseFirst;
  if constexpr (!kFuseFirst) {
    {
      // Lane tid's k-th element is e = tid + k NT: the same column, NT / ncol rows further down, so
      // one per-lane offset and a uniform step address every load and every LDS write.
      constexpr int U = 4096 / NT;  // loads in flight per lane before the first LDS write
      const int e_t = min(tid, n_elem - 1);
      const int c_t = e_t & (ncol - 1);
      const int pos_t = c_t < na ? un.a0 + c_t : un.b0 + (c_t - na);
      const cx<T>* src = Zs + (pos_t / kSlab) * sp + (e_t >> lncol) * kSlab + pos_t % kSlab;
      const int src_step = (NT >> lncol) * kSlab;
      cx<T>* dst = buf + (tid >> lncol) * rs + (tid & (ncol - 1));
      const int dst_step = (NT >> lncol) * rs;
      for (int e0 = tid, k0 = 0; e0 < n_elem; e0 += NT * U, k0 += U) {
        cx<T> r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = stream_load<2>(&src[e0 + u * NT < n_elem ? (k0 + u) * src_step : 0]);  // never branch around a load
#pragma unroll
        for (int u = 0; u < U; ++u) {
          MS_PIN(r[u].x);
          MS_PIN(r[u].y);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (e0 + u * NT < n_elem) dst[(k0 + u) * dst_step] = r[u];
      }
    }
    lds_barrier();
    MS_STAMP(1, 1);
    if constexpr (MIXED)
      mixed_forward(buf, rs, lncol, planS, twN, tid, NT);
    else
      fft_forward<T, MAXR>(buf, addr, ncol, planS, tw, tid, NT);
  } else {
    // the first radix pass takes the columns straight from the slabs: lanes of one row read its 2C
    // columns (two runs of 128 or 64 bytes), and nothing is written to LDS that the pass would only read back
    MS_STAMP(1, 1);
    auto from_slabs = [&](int c, int q) {
      const int pos = c < na ? un.a0 + c : un.b0 + (c - na);
      return stream_load<2>(&Zs[(pos / kSlab) * sp + q * kSlab + pos % kSlab]);
    };
    if constexpr (MIXED)
      mixed_forward_from(buf, rs, lncol, planS, twN, tid, NT, true, from_slabs);
    else
      fft_forward_from<T, MAXR>(buf, addr, ncol, planS, tw, tid, NT, true, from_slabs);
  }
  MS_STAMP(1, 2);
  const bool partner_in_a = mp >= un.a0 && mp < un.a0 + na;
  const int pc = partner_in_a ? (mp - un.a0) : (na + mp - un.b0);
  for (int e = tid; e < S * na; e += NT) {
    const int q = e >> lna, c = e & (na - 1);
    const int q2 = mposS[q];
    if (partner_in_a && (q2 * rs + pc) < (q * rs + c)) continue;  // the partner cell's owner writes both
    const cx<T> z = buf[q * rs + c];
    const cx<T> zm = buf[q2 * rs + pc];
    // z = A + iB with A, B the spectra of the two real strips:
    // A = (z + conj(zm)) / 2,  B = (z - conj(zm)) / (2i)
    T ax = (T)0.5 * (z.x + zm.x), ay = (T)0.5 * (z.y - zm.y);
    T bx = (T)0.5 * (z.y + zm.y), by = (T)-0.5 * (z.x - zm.x);
    if (exact && (q2 * rs + pc) == (q * rs + c)) exact_real_bin(ax, bx);  // its own partner: A = Re z, B = Im z
    T fx = ax * bx + ay * by;  // A * conj(B)
    T fy = ay * bx - ax * by;
    const T inv = (T)1 / (cabs_safe<T>(fx, fy) + eps);
    fx *= inv;
    fy *= inv;
    buf[q * rs + c] = mk<T>(fx, fy);
    if (partner_in_a && (q2 * rs + pc) != (q * rs + c)) buf[q2 * rs + pc] = mk<T>(fx, -fy);
  }
  lds_barrier();
  MS_STAMP(1, 3);
  if (keep_r != nullptr) {
    // the subpixel stage wants the normalised spectrum itself: it replaces this unit's own
    // columns of the slab buffer, which no other workgroup reads, with the rows reordered
    // to ascending signed frequency -S/2 .. S/2-1
    cx<T>* Rs = keep_r + (int64_t)blockIdx.y * slab_pair_elems(S, L);
    for (int e = tid; e < S * na; e += NT) {
      const int q = e >> lna, c = e & (na - 1);
      const int pos = un.a0 + c;
      Rs[(pos / kSlab) * sp + rowS[q] * kSlab + pos % kSlab] = buf[q * rs + c];
    }
    if constexpr (MIXED) {
      if ((S & 1) && tid < na) {
        const int pos = un.a0 + tid;
        Rs[(pos / kSlab) * sp + pos % kSlab] = mk<T>((T)0, (T)0);
      }
    }
    lds_barrier();  // the inverse transform's first pass overwrites cells other lanes copy out here
  }
  // the intermediate spectrum holds the stored columns only, packed in ascending position
  // order: the last pass then reads whole rows contiguously
  if constexpr (!kFuseLast) {
    if constexpr (MIXED)
      mixed_inverse(buf, rs, lna, planS, twN, tid, NT);
    else
      fft_inverse<T, MAXR>(buf, addr, na, planS, tw, tid, NT);
    MS_STAMP(1, 4);
    for (int e = tid; e < S * na; e += NT) {
      const int s = e >> lna, c = e & (na - 1);
      Z[(int64_t)s * Lp + col0 + c] = buf[s * rs + c];
    }
  } else {
    // the last radix pass stores its results itself (the na columns of a row are one run of 128 or 64 bytes)
    auto to_spectrum = [&](int c, int q, cx<T> v) { Z[(int64_t)q * Lp + col0 + c] = v; };
    if constexpr (MIXED)
      mixed_inverse_to(buf, rs, lna, planS, twN, tid, NT, to_spectrum);
    else
      fft_inverse_to<T, MAXR>(buf, addr, na, planS, tw, tid, NT, to_spectrum);
    MS_STAMP(1, 4);
  }
  MS_STAMP(1, 5);
}

// The column pass for complex128 columns of S = 128 or 256 rows with every radix pass in registers (the scheme of
// fft_reg.hpp along the short axis).  A workgroup takes NC stored columns and the NC columns of their negated
// frequencies: lanes [0, NC S / 8) take the stored columns, the rest the others, lane (c, q) = (lane % NC, lane / NC)
// holding 8 rows of column c.  In memory the 8 columns of a unit are ONE contiguous [S][8] block of the slab buffer,
// so a wave's loads cover whole rows (NC = 8) or half rows (NC = 4).  The passes (S / 64, 8, 8) exchange through LDS
// (row r of a half at slot r + r / 8: conflict free for the row sets of every pass); the forward transform's last
// pass leaves a lane the 8 storage positions that the inverse transform's first pass starts from, so the cross-power
// stage works on registers and only reads the partner cells from LDS.  The second half's waves leave once their
// transform is published.
// NC = 8: one unit per workgroup, and the units whose stored columns hold their own negated frequencies (nb = 0:
// the second half repeats the first).  NC = 4: HALF a paired unit per workgroup - the mirror of stored column c of a
// unit is column 7 - c of its partner block, so columns 0-3 go with the partner's 4-7 and the reverse; four
// workgroups of 256 lanes share a CU instead of two of 512; the two halves of a unit, which share every 128-byte row
// of its slabs, are dealt to the same XCD (grid.x a multiple of 16: block x -> unit (x / 16) 8 + x % 8, half x / 8 % 2).
// Measured 2 % SLOWER than whole units on the 2048 x 256 strips (2.37 against 2.32 ms per 760 pairs), so it only runs
// when MISTITCH_CPS_HALF_UNITS is set in the environment (an experiment switch).
// grid (n_units or 16 ceil(n_units / 8), pairs_in_pass), 2 NC S / 8 lanes.
template <int S, int NC>
__global__ __launch_bounds__(2 * NC * S / 8, NC == 8 ? 4 : 4) void k_fft_short_cps_reg(
    const PairDesc* desc, const int8_t* status, int pair0, int L, int want_transposed, const double2* __restrict__ roots,
    const uint16_t* __restrict__ mpos_s, const uint16_t* __restrict__ pos2freq_s, const uint16_t* __restrict__ mpos_l,
    const CpsUnit* __restrict__ units, int n_units, const double2* slabs, double2* spectrum, double eps, double2* keep_r,
    const uint16_t* __restrict__ pos2stored, int n_stored, int exact) {
  constexpr int Q = S / 8;            // lanes per column
  constexpr int R0 = S / 64;          // radix of the first pass
  constexpr int NB = 8 / R0;          // its butterflies per lane
  constexpr int kRows = S + S / 8;    // slots of a half
  constexpr int kHalfLanes = NC * Q;
  static_assert((S == 128 || S == 256) && (NC == 4 || NC == 8) && kHalfLanes % 64 == 0, "passes (S / 64, 8, 8), whole waves per half");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  const int8_t pair_status = status[p];
  const int d_S = desc[p].S, d_L = desc[p].L, d_transposed = desc[p].transposed;
  int ui = blockIdx.x, hh = 0;
  if constexpr (NC == 4) {
    ui = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7);
    hh = ((int)blockIdx.x >> 3) & 1;
  }
  if (ui >= n_units) return;
  const CpsUnit un = units[ui];
  const int tid = threadIdx.x;
  const int half = tid >= kHalfLanes ? 1 : 0, lt = tid - half * kHalfLanes;
  const int c = lt & (NC - 1), q = lt / NC;
  const bool single = NC == 8 && un.nb == 0;  // the stored columns hold their own negated frequencies: the second half repeats them
  const int a_base = un.a0 + (NC == 4 ? 4 * hh : 0);
  const int b_base = single ? a_base : un.b0 + (NC == 4 ? 4 * (1 - hh) : 0);
  const int pos = (half == 0 ? a_base : b_base) + c;
  const int64_t sp = slab_pitch(S);
  const double2* src = slabs + (int64_t)blockIdx.y * slab_pair_elems(S, L) + (pos / kSlab) * sp + pos % kSlab;
  double2 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = stream_load<2>(&src[(q + Q * j) * kSlab]);
  double2 tw0[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) tw0[k] = roots[q + k * Q];
  const double2 tw1 = roots[(q & 7) * (S / 64)];
  const int mp = mpos_l[a_base + c];
  const uint4 mrow4 = *reinterpret_cast<const uint4*>(mpos_s + 8 * q);       // partner rows of rows 8 q .. 8 q + 7
  const int col0 = pos2stored[a_base];
  if (pair_status != MISTITCH_PAIR_OK) return;
  if (d_S != S || d_L != L || (want_transposed >= 0 && d_transposed != want_transposed)) return;
  double2* mine = buf + half * (kRows * NC);
  // first pass: rows q + Q j
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    double2 x[R0];
#pragma unroll
    for (int m = 0; m < R0; ++m) x[m] = v[k + NB * m];
    RegFft<512>::butterfly<R0, false>(x, tw0[k], true);
#pragma unroll
    for (int m = 0; m < R0; ++m) v[k + NB * m] = x[m];
  }
  auto slot = [](int r) { return r + (r >> 3); };
  double2* at_a = mine + slot(q) * NC + c;                          // rows q + Q j
  double2* at_b = mine + slot((q >> 3) * 64 + (q & 7)) * NC + c;    // rows (q / 8) 64 + q % 8 + 8 m
  double2* at_c = mine + 9 * q * NC + c;                            // rows 8 q + m
#pragma unroll
  for (int j = 0; j < 8; ++j) at_a[j * (Q + Q / 8) * NC] = v[j];
  lds_barrier();
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = at_b[9 * m * NC];
  RegFft<512>::butterfly<8, false>(v, tw1, true);
#pragma unroll
  for (int m = 0; m < 8; ++m) at_b[9 * m * NC] = v[m];  // in place: the slots this lane has just read
  lds_barrier();
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = at_c[m * NC];
  RegFft<512>::butterfly<8, false>(v, mk<double>(1.0, 0.0), false);
  if (exact && single && half == 0 && mp == a_base + c) {
    // a cell that is its own partner holds A = Re z, B = Im z, integers (see exact_real_bin): rounded here, before
    // the copy that the partner read below fetches, and outside the cross-power loop, which has no register to spare
    const unsigned own = 8u * (unsigned)q;
    if ((mrow4.x & 0xffffu) == own) exact_real_bin(v[0].x, v[0].y);
    if ((mrow4.x >> 16) == own + 1) exact_real_bin(v[1].x, v[1].y);
    if ((mrow4.y & 0xffffu) == own + 2) exact_real_bin(v[2].x, v[2].y);
    if ((mrow4.y >> 16) == own + 3) exact_real_bin(v[3].x, v[3].y);
    if ((mrow4.z & 0xffffu) == own + 4) exact_real_bin(v[4].x, v[4].y);
    if ((mrow4.z >> 16) == own + 5) exact_real_bin(v[5].x, v[5].y);
    if ((mrow4.w & 0xffffu) == own + 6) exact_real_bin(v[6].x, v[6].y);
    if ((mrow4.w >> 16) == own + 7) exact_real_bin(v[7].x, v[7].y);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) at_c[m * NC] = v[m];
  lds_barrier();
  if (half == 1) return;  // whole waves: the barriers below count the surviving ones
  // normalised cross-power spectrum of the two strips packed in z = A + i B (see k_fft_short_cps)
  const double2* partner = buf + (single ? 0 : kRows * NC) + (mp - b_base);
  const unsigned mrow[8] = {mrow4.x & 0xffffu, mrow4.x >> 16, mrow4.y & 0xffffu, mrow4.y >> 16,
                            mrow4.z & 0xffffu, mrow4.z >> 16, mrow4.w & 0xffffu, mrow4.w >> 16};
#pragma unroll
  for (int h = 0; h < 8; h += 4) {  // four cells at a time: their partner reads in flight together
    double2 zm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) zm[m] = partner[slot((int)mrow[h + m]) * NC];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double2 z = v[h + m];
      const double ax = 0.5 * (z.x + zm[m].x), ay = 0.5 * (z.y - zm[m].y);
      const double bx = 0.5 * (z.y + zm[m].y), by = -0.5 * (z.x - zm[m].x);
      double fx = ax * bx + ay * by;  // A * conj(B)
      double fy = ay * bx - ax * by;
      const double inv = 1.0 / (cabs_safe<double>(fx, fy) + eps);
      v[h + m] = mk<double>(fx * inv, fy * inv);
    }
  }
  if (keep_r != nullptr) {
    // the subpixel stage's copy replaces this unit's own slab, rows reordered to ascending signed frequency
    const uint4 frow4 = *reinterpret_cast<const uint4*>(pos2freq_s + 8 * q);  // frequencies of rows 8 q .. 8 q + 7
    const unsigned frow[8] = {frow4.x & 0xffffu, frow4.x >> 16, frow4.y & 0xffffu, frow4.y >> 16,
                              frow4.z & 0xffffu, frow4.z >> 16, frow4.w & 0xffffu, frow4.w >> 16};
    double2* Rs = keep_r + (int64_t)blockIdx.y * slab_pair_elems(S, L) + (pos / kSlab) * sp + pos % kSlab;
#pragma unroll
    for (int m = 0; m < 8; ++m) stream_store<256>(&Rs[(int)((frow[m] + S / 2) & (S - 1)) * kSlab], v[m]);
  }
  // inverse transform of the stored columns: the first adjoint pass starts from the rows a lane holds
  RegFft<512>::butterfly<8, true>(v, mk<double>(1.0, 0.0), false);
  if (single) lds_barrier();  // the partner cells lie in this half: every lane has to have read them (workgroup uniform)
#pragma unroll
  for (int m = 0; m < 8; ++m) at_c[m * NC] = v[m];
  lds_barrier();
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = at_b[9 * m * NC];
  RegFft<512>::butterfly<8, true>(v, tw1, true);
#pragma unroll
  for (int m = 0; m < 8; ++m) at_b[9 * m * NC] = v[m];  // in place
  lds_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = at_a[j * (Q + Q / 8) * NC];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    double2 x[R0];
#pragma unroll
    for (int m = 0; m < R0; ++m) x[m] = v[k + NB * m];
    RegFft<512>::butterfly<R0, true>(x, tw0[k], true);
#pragma unroll
    for (int m = 0; m < R0; ++m) v[k + NB * m] = x[m];
  }
  // natural order out: rows q + Q j of the packed intermediate spectrum, NC columns (128 or 64 bytes) per row
  const int Lp = spectrum_pitch<double>(n_stored);
  double2* Z = spectrum + (int64_t)blockIdx.y * S * Lp + col0 + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) stream_store<4>(&Z[(int64_t)(q + Q * j) * Lp], v[j]);
}

// Inverse FFT along the long axis, two lines per complex transform, then the
// block's k largest values.  grid (S / (2 NF), pairs_in_pass).
template <typename T>
__global__ __launch_bounds__(kThreads) void k_fft_long_inv(const PairDesc* desc, const int8_t* status,
                                                           int pair0, int S, int L, int want_transposed,
                                                           FftPlanDev planL, const uint16_t* __restrict__ stored_pos,
                                                           const uint16_t* __restrict__ mirror_fill, int n_stored,
                                                           int NF, const cx<T>* spectrum, int k, Peak* partial,
                                                           int partial_stride, double* pcm_out, Peak* partial_abs,
                                                           int abs_stride, int nblk, int seq) {
  // nblk: blocks of 2 NF lines a strip has (= entries of a pair's partial lists); seq: how many of them, one after
  // the other, this workgroup transforms (more than one only where the values are ranked in the last radix pass:
  // the per-lane ranking then runs across all of them and the workgroup reduces once; its other list entries are
  // written empty).  grid (ceil(nblk / seq), pairs_in_pass).
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cx<T>* buf = reinterpret_cast<cx<T>*>(smem_raw);
  __shared__ Peak red[2 * kThreads / 64];
  const int p = pair0 + blockIdx.y;
  // the pair's status and descriptor are looked up here but tested only once the spectrum loads are on their
  // way (the workspace rows exist whatever the pair's state): one memory round trip in front of the transform
  const int8_t pair_status = status[p];
  const PairDesc d = desc[p];
  const int tid = threadIdx.x;
  LineAddr<T> addr{LineAddr<T>::padded(L) + 1};
  const Twiddles<T> tw = load_twiddles<T>(buf + NF * addr.fstride, planL, tid, kThreads);
  const int blk0 = blockIdx.x * seq;
  int s0 = blk0 * NF * 2;
  const int lL = 31 - __builtin_clz(L);  // a power of two on this path
  const int Lp = spectrum_pitch<T>(n_stored);
  const cx<T>* Z = spectrum + (int64_t)blockIdx.y * S * Lp;
  MS_STAMP(2, 0);
  // lines s0 .. s0 + 2 NF into LDS; false (for every lane alike) when the pair is not this launch's to transform
  auto load_lines = [&]() -> bool {
    // Only the stored half of every line is read; the other half is its conjugate
    // mirror and is written to LDS by the lane that loaded the stored value.
    constexpr int U = 8;  // two loads per element: 16 in flight per lane
    for (int e0 = tid; e0 < NF * n_stored; e0 += kThreads * U) {
      cx<T> g1[U], g2[U];
      // the positions the values go to are looked up with the values, not after them: one memory round trip, not two
      uint16_t at_pos[U], at_mir[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = min(e0 + u * kThreads, NF * n_stored - 1);  // clamp, do not branch around the loads
        const int f = NF == 1 ? 0 : e / n_stored, i = e - f * n_stored;
        // lines past the last one (odd S, or a last block that is not full) read the last line and count as zero
        g1[u] = stream_load<8>(&Z[(int64_t)min(s0 + 2 * f, S - 1) * Lp + i]);
        g2[u] = stream_load<8>(&Z[(int64_t)min(s0 + 2 * f + 1, S - 1) * Lp + i]);
        at_pos[u] = stored_pos[i];
        at_mir[u] = mirror_fill[i];
      }
      if (pair_status != MISTITCH_PAIR_OK || d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed))
        return false;  // the same for every lane of the workgroup
#pragma unroll
      for (int u = 0; u < U; ++u) {
        MS_PIN(g1[u].x); MS_PIN(g1[u].y);
        MS_PIN(g2[u].x); MS_PIN(g2[u].y);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        if (e < NF * n_stored) {
          const int f = NF == 1 ? 0 : e / n_stored;
          const int pos = at_pos[u], mir = at_mir[u];
          if (s0 + 2 * f >= S) g1[u] = mk<T>((T)0, (T)0);
          if (s0 + 2 * f + 1 >= S) g2[u] = mk<T>((T)0, (T)0);
          buf[addr(f, pos)] = mk<T>(g1[u].x - g2[u].y, g1[u].y + g2[u].x);  // g1 + i g2
          if (mir != 0xffff) buf[addr(f, mir)] = mk<T>(g1[u].x + g2[u].y, g2[u].x - g1[u].y);  // conj(g1) + i conj(g2)
        }
      }
    }
    return true;
  };
  if (!load_lines()) return;
  __syncthreads();
  MS_STAMP(2, 1);
  // The production case (two peaks, no PCM plane wanted) ranks the values as the last radix pass produces
  // them, in registers: they are never written back to LDS and read again.
  const bool rank_in_last_pass = k > 0 && k <= 2 && pcm_out == nullptr;
  if (!rank_in_last_pass) fft_inverse<T, kLongMaxRadix<T>>(buf, addr, NF, planL, tw, tid, kThreads);
  MS_STAMP(2, 2);
  const double scale = 1.0 / ((double)S * (double)L);
  if (pcm_out != nullptr) {
    double* out = pcm_out + (int64_t)blockIdx.y * S * L;
    for (int e = tid; e < NF * L; e += kThreads) {
      const int f = e >> lL, l = e & (L - 1);
      const cx<T> v = buf[addr(f, l)];
      const int s = s0 + 2 * f;
      if (d.transposed) {
        if (s < S) out[(int64_t)l * S + s] = (double)v.x * scale;
        if (s + 1 < S) out[(int64_t)l * S + s + 1] = (double)v.y * scale;
      } else {
        if (s < S) out[(int64_t)s * L + l] = (double)v.x * scale;
        if (s + 1 < S) out[(int64_t)(s + 1) * L + l] = (double)v.y * scale;
      }
    }
  }
  if (k <= 0) return;
  T* flat = reinterpret_cast<T*>(buf);
  const T neg_inf = -std::numeric_limits<T>::infinity();
  if (k <= 2) {
    // The production case: one scan keeps each lane's two best values, then k argmax
    // rounds in which a lane whose best was taken offers its second best.
    // Values are compared unscaled where the scale is a power of two (nothing then changes order or
    // creates ties), scaled otherwise, and carried as (value, key) only.
    const bool exact_scale = (S & (S - 1)) == 0;
    const T cmp_scale = exact_scale ? (T)1 : (T)scale;
    const double out_scale = exact_scale ? scale : 1.0;
    T v1 = neg_inf, v2 = neg_inf;
    int k1 = -1, k2 = -1;
    Peak ba{-INFINITY, -1, -1};  // largest magnitude, first in row-major order (subpixel stage only)
    // branch-free: one pass of compares and selects per value (`have` false: the value does not exist)
    auto offer = [&](T v, int key, bool have) {
      const bool gt1 = have & ((v > v1) | ((v == v1) & (key > k1)));
      const bool gt2 = have & ((v > v2) | ((v == v2) & (key > k2)));
      v2 = gt1 ? v1 : (gt2 ? v : v2);
      k2 = gt1 ? k1 : (gt2 ? key : k2);
      v1 = gt1 ? v : v1;
      k1 = gt1 ? key : k1;
    };
    const bool want_abs = partial_abs != nullptr;
    auto take = [&](int f, int l, cx<T> v) {
      const int s = s0 + 2 * f;
      const int key0 = d.transposed ? l * S + s : s * L + l;
      const int key1 = d.transposed ? key0 + 1 : key0 + L;
      const bool have0 = s < S, have1 = s + 1 < S;  // an odd S leaves the last transform half empty
      offer(v.x * cmp_scale, key0, have0);
      offer(v.y * cmp_scale, key1, have1);
      if (want_abs) {
        const double a0v = fabs((double)v.x * scale), a1v = fabs((double)v.y * scale);
        const bool t0 = have0 & first_better(a0v, key0, ba.val, ba.key);
        ba.val = t0 ? a0v : ba.val;
        ba.key = t0 ? key0 : ba.key;
        const bool t1 = have1 & first_better(a1v, key1, ba.val, ba.key);
        ba.val = t1 ? a1v : ba.val;
        ba.key = t1 ? key1 : ba.key;
      }
    };
    if (rank_in_last_pass) {
      for (int it = 0;;) {
        fft_inverse_to<T, kLongMaxRadix<T>>(buf, addr, NF, planL, tw, tid, kThreads, take);
        s0 += NF * 2;
        if (++it >= seq || s0 >= S) break;  // the same for every lane
        __syncthreads();                    // the last pass has read the lines that the next ones replace
        load_lines();
        __syncthreads();
      }
    } else {
      for (int e = tid; e < NF * L; e += kThreads) {
        const int f = e >> lL, l = e & (L - 1);
        take(f, l, buf[addr(f, l)]);
      }
    }
    ba.loc = ba.key >= 0 ? 0 : -1;
    Peak b1{k1 >= 0 ? (double)v1 * out_scale : -INFINITY, k1, k1}, b2{k2 >= 0 ? (double)v2 * out_scale : -INFINITY, k2, k2};
    const Peak none{-INFINITY, -1, -1};
    if (partial_abs != nullptr) {
      const Peak best = block_argmax_first(ba, red);
      if (tid == 0) partial_abs[(int64_t)blockIdx.y * abs_stride + blk0] = best;
      if (tid > 0 && tid < seq && blk0 + tid < nblk) partial_abs[(int64_t)blockIdx.y * abs_stride + blk0 + tid] = none;
      __syncthreads();
    }
    // every wave's two best (the lane whose best was the wave's first offers its second), then one
    // barrier and a merge of the 2 x 4 entries by one thread: one barrier instead of two per round
    const Peak w1 = wave_best_peak(b1, peak_better);
    if (w1.loc >= 0 && w1.loc == b1.loc) b1 = b2;
    const Peak w2 = k > 1 ? wave_best_peak(b1, peak_better) : none;
    if ((tid & 63) == 0) {
      red[2 * (tid >> 6)] = w1;
      red[2 * (tid >> 6) + 1] = w2;
    }
    __syncthreads();
    if (tid == 0) {
      Peak t1 = none, t2 = none;
#pragma unroll
      for (int e = 0; e < 2 * kThreads / 64; ++e) {
        const Peak o = red[e];
        if (peak_better(o.val, o.key, t1.val, t1.key)) {
          t2 = t1;
          t1 = o;
        } else if (peak_better(o.val, o.key, t2.val, t2.key)) {
          t2 = o;
        }
      }
      Peak* dst = partial + (int64_t)blockIdx.y * partial_stride + blk0 * k;
      dst[0] = t1;
      if (k > 1) dst[1] = t2;
    } else if (tid < seq && blk0 + tid < nblk) {  // the list entries of the other blocks this workgroup covered
      Peak* dst = partial + (int64_t)blockIdx.y * partial_stride + (blk0 + tid) * k;
      dst[0] = none;
      if (k > 1) dst[1] = none;
    }
    MS_STAMP(2, 3);
    return;
  }
  if (partial_abs != nullptr) {  // largest magnitude, first in row-major order (subpixel stage only)
    Peak ba{-INFINITY, -1, -1};
    for (int e = tid; e < NF * L; e += kThreads) {
      const int f = e >> lL, l = e & (L - 1);
      const cx<T> v = buf[addr(f, l)];
      const int s = s0 + 2 * f;
      const int key0 = d.transposed ? l * S + s : s * L + l;
      const int key1 = d.transposed ? key0 + 1 : key0 + L;
      const double a0v = fabs((double)v.x * scale), a1v = fabs((double)v.y * scale);
      if (s < S && first_better(a0v, key0, ba.val, ba.key)) ba = Peak{a0v, key0, 0};
      if (s + 1 < S && first_better(a1v, key1, ba.val, ba.key)) ba = Peak{a1v, key1, 0};
    }
    const Peak best = block_argmax_first(ba, red);
    if (tid == 0) partial_abs[(int64_t)blockIdx.y * abs_stride + blockIdx.x] = best;
    __syncthreads();
  }
  // k rounds of block argmax; a taken element is overwritten with -inf.
  for (int round = 0; round < k; ++round) {
    Peak mine{-INFINITY, -1, -1};
    for (int e = tid; e < NF * L; e += kThreads) {
      const int f = e / L, l = e - f * L;
      const int a0 = addr(f, l);
      const cx<T> v = buf[a0];
      const int s = s0 + 2 * f;
      const int key0 = d.transposed ? l * S + s : s * L + l;
      const int key1 = d.transposed ? key0 + 1 : key0 + L;
      const double v0 = (double)v.x * scale, v1 = (double)v.y * scale;
      if (s < S && v.x != neg_inf && peak_better(v0, key0, mine.val, mine.key)) mine = Peak{v0, key0, 2 * a0};
      if (s + 1 < S && v.y != neg_inf && peak_better(v1, key1, mine.val, mine.key)) mine = Peak{v1, key1, 2 * a0 + 1};
    }
    Peak best = block_argmax(mine, red);
    if (tid == 0) {
      partial[(int64_t)blockIdx.y * partial_stride + blockIdx.x * k + round] = best;
      if (best.loc >= 0) flat[best.loc] = neg_inf;
    }
    __syncthreads();
  }
  MS_STAMP(2, 3);
}

// The production case of that pass (two peaks at most, no correlation plane wanted) for complex128 lines of
// N = 512 .. 4096 points on the register-resident engine: the stored half of two spectrum rows is spread over the
// line in LDS as before, the adjoint passes run in registers, and the values are ranked where the last pass
// leaves them.  grid (ceil(nblk / seq), pairs_in_pass), N / 8 lanes, blocks of two lines (nblk per strip).
// WIDE: more than two peaks (the ten-peak candidate set), one block of lines per workgroup.  Every wave finds the
// ceil(k / waves) largest of its lanes' maxima; the smallest of those over the waves is a bound T that at least k
// values of the block reach, so the k largest are all among the values >= T: those (a dozen or two) are appended to a
// list in LDS and one wave ranks them.  A block with more than kInvListCap values >= T (massive ties) is ranked by k
// rounds of a block-wide argmax over the registers instead.
#ifndef MS_INV_LIST_CAP
#define MS_INV_LIST_CAP 128
#endif
constexpr int kInvListCap = MS_INV_LIST_CAP;
static_assert(kInvListCap <= 128 && kInvListCap >= 1, "one wave ranks the list, two entries per lane");

// Largest value of a double over the wave, the same in every lane.
__device__ __forceinline__ double wave_max_f64(double v) {
  auto step = [&](unsigned long long o) { v = fmax(v, __longlong_as_double((long long)o)); };
  step(dpp_u64<kDppXor1>((unsigned long long)__double_as_longlong(v)));
  step(dpp_u64<kDppXor2>((unsigned long long)__double_as_longlong(v)));
  step(dpp_u64<kDppHalfMirror>((unsigned long long)__double_as_longlong(v)));
  step(dpp_u64<kDppMirror>((unsigned long long)__double_as_longlong(v)));
  double m = v;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)bits, 16 * row), hi = __builtin_amdgcn_readlane((unsigned)(bits >> 32), 16 * row);
    m = fmax(m, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
  }
  return m;
}

template <int N, bool ABS, bool WIDE = false>
__global__ __launch_bounds__(N / 8, 4) void k_fft_long_inv_reg(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                            int want_transposed, const double2* __restrict__ roots,
                                                            const uint16_t* __restrict__ stored_pos,
                                                            const uint16_t* __restrict__ mirror_fill, int n_stored,
                                                            const double2* spectrum, int k, Peak* partial,
                                                            int partial_stride, Peak* partial_abs, int abs_stride,
                                                            int nblk, int seq) {
  using F = RegFft<N>;
  constexpr int NT = N / 8, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  __shared__ Peak red[2 * NW];
  __shared__ Peak wide_list[WIDE ? kInvListCap : 1];
  __shared__ double wide_thr[NW];
  __shared__ int wide_n;
  const int p = pair0 + blockIdx.y;
  const int8_t pair_status = status[p];
  const PairDesc d = desc[p];
  const int tid = threadIdx.x;
  typename F::Tw tw = F::load_twiddles(roots, tid);
  const int blk0 = blockIdx.x * seq;
  int s0 = blk0 * 2;
  const int Lp = spectrum_pitch<double>(n_stored);
  const double2* Z = spectrum + (int64_t)blockIdx.y * S * Lp;
  // lines s0, s0 + 1 into LDS as g1 + i g2; false (for every lane alike) when the pair is not this launch's
  auto load_lines = [&]() -> bool {
    // the stored half is N / 2 + 9 values: five per lane cover it in ONE round of loads (four would leave nine values
    // to a second round, a second trip to memory in front of every transform)
#ifndef MS_INV_LOADS
#define MS_INV_LOADS 5
#endif
    constexpr int U = MS_INV_LOADS;
    const double2* z1 = Z + (int64_t)min(s0, S - 1) * Lp;      // lines past the last one read it and count as zero
    const double2* z2 = Z + (int64_t)min(s0 + 1, S - 1) * Lp;
    for (int e0 = tid; e0 < n_stored; e0 += NT * U) {
      double2 g1[U], g2[U];
      uint16_t at_pos[U], at_mir[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(e0 + u * NT, n_stored - 1);  // clamp, do not branch around the loads
        g1[u] = stream_load<8>(&z1[i]);
        g2[u] = stream_load<8>(&z2[i]);
        at_pos[u] = stored_pos[i];
        at_mir[u] = mirror_fill[i];
      }
      if (pair_status != MISTITCH_PAIR_OK || d.S != S || d.L != N || (want_transposed >= 0 && d.transposed != want_transposed))
        return false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        MS_PIN(g1[u].x); MS_PIN(g1[u].y);
        MS_PIN(g2[u].x); MS_PIN(g2[u].y);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e0 + u * NT < n_stored) {
          const int pos = at_pos[u], mir = at_mir[u];
          if (s0 >= S) g1[u] = mk<double>(0.0, 0.0);
          if (s0 + 1 >= S) g2[u] = mk<double>(0.0, 0.0);
          buf[F::slot(pos)] = mk<double>(g1[u].x - g2[u].y, g1[u].y + g2[u].x);  // g1 + i g2
          if (mir != 0xffff) buf[F::slot(mir)] = mk<double>(g1[u].x + g2[u].y, g2[u].x - g1[u].y);  // conj(g1) + i conj(g2)
        }
      }
    }
    return true;
  };
  if (!load_lines()) return;
  __syncthreads();
  const double scale = 1.0 / ((double)S * (double)N);
  // values are compared unscaled where the scale is a power of two, scaled otherwise (see k_fft_long_inv)
  const bool exact_scale = (S & (S - 1)) == 0;
  const double cmp_scale = exact_scale ? 1.0 : scale;
  const double out_scale = exact_scale ? scale : 1.0;
  double v1 = -INFINITY, v2 = -INFINITY;
  int k1 = -1, k2 = -1;
  Peak ba{-INFINITY, -1, -1};  // largest magnitude, first in row-major order (subpixel stage only)
  auto offer = [&](double v, int key, bool have) {
    const bool gt1 = have & ((v > v1) | ((v == v1) & (key > k1)));
    const bool gt2 = have & ((v > v2) | ((v == v2) & (key > k2)));
    v2 = gt1 ? v1 : (gt2 ? v : v2);
    k2 = gt1 ? k1 : (gt2 ? key : k2);
    v1 = gt1 ? v : v1;
    k1 = gt1 ? key : k1;
  };
  constexpr bool want_abs = ABS;  // (partial_abs is given exactly then)
  for (int it = 0;;) {
    double2 v[8];
    // the twiddle bases are pinned per trip: their powers are then formed where they are used, not hoisted out of
    // the loop and kept (80 registers) across it
#pragma unroll
    for (int kk = 0; kk < F::NB; ++kk) {
      MS_PIN(tw.p0[kk].x);
      MS_PIN(tw.p0[kk].y);
    }
    MS_PIN(tw.p1.x); MS_PIN(tw.p1.y);
    MS_PIN(tw.p2.x); MS_PIN(tw.p2.y);
#if defined(MS_INV_ABLATE) && MS_INV_ABLATE == 1  /* measurement build: no transform */
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = buf[tid + NT * j];
#else
    F::inverse(v, buf, tw, tid);
#endif
    const bool have0 = s0 < S, have1 = s0 + 1 < S;  // an odd S leaves the last transform half empty
    if constexpr (WIDE) {  // (seq == 1: this is the workgroup's only block)
      const int lane = tid & 63, wave = tid >> 6;
      auto val_of = [&](int e) { return (e & 1 ? v[e >> 1].y : v[e >> 1].x) * cmp_scale; };
      auto key_of = [&](int e) {
        const int l = tid + NT * (e >> 1);
        const int key0 = d.transposed ? l * S + s0 : s0 * N + l;
        return e & 1 ? (d.transposed ? key0 + 1 : key0 + N) : key0;
      };
      double lane_max = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) lane_max = ((e & 1) ? have1 : have0) ? fmax(lane_max, val_of(e)) : lane_max;
      const int per_wave = (k + NW - 1) / NW;
      double cur = lane_max, thr = -INFINITY;
      for (int r = 0; r < per_wave; ++r) {  // the wave's per_wave-th largest lane maximum
        thr = wave_max_f64(cur);
        const unsigned long long at = __ballot(cur == thr);
        if (at != 0ull && lane == __builtin_ctzll(at)) cur = -INFINITY;
      }
      if (lane == 0) wide_thr[wave] = thr;
      if (tid == 0) wide_n = 0;
      __syncthreads();
      double bound = wide_thr[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) bound = fmin(bound, wide_thr[w]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const double x = val_of(e);
        if (((e & 1) ? have1 : have0) && x >= bound) {
          const int at = atomicAdd(&wide_n, 1);
          if (at < kInvListCap) wide_list[at] = Peak{x, key_of(e), at};
        }
      }
      __syncthreads();
      const int n_list = wide_n;
      const Peak none{-INFINITY, -1, -1};
      Peak* dst = partial + (int64_t)blockIdx.y * partial_stride + blk0 * k;
      if (n_list <= kInvListCap) {  // (uniform) one wave ranks the list, two entries per lane
        if (wave == 0) {
          Peak e0 = lane < n_list ? wide_list[lane] : none;
          Peak e1 = lane + 64 < n_list ? wide_list[lane + 64] : none;
          for (int round = 0; round < k; ++round) {
            const Peak mine = peak_better(e1.val, e1.key, e0.val, e0.key) ? e1 : e0;
            const Peak w = wave_best_peak(mine, peak_better);
            if (lane == 0) dst[round] = w.key >= 0 ? Peak{w.val * out_scale, w.key, w.key} : none;
            if (w.key >= 0 && e0.loc == w.loc) e0 = none;
            if (w.key >= 0 && e1.loc == w.loc) e1 = none;
          }
        }
      } else {  // (uniform) too many values reach the bound: k rounds of a block-wide argmax over the registers
        unsigned taken = 0;
        for (int round = 0; round < k; ++round) {
          Peak mine = none;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const double x = val_of(e);
            const int key = key_of(e);
            const bool ok = ((e & 1) ? have1 : have0) & !((taken >> e) & 1u) & peak_better(x, key, mine.val, mine.key);
            mine.val = ok ? x : mine.val;
            mine.key = ok ? key : mine.key;
            mine.loc = ok ? tid * 16 + e : mine.loc;
          }
          const Peak w = wave_best_peak(mine, peak_better);
          __syncthreads();
          if (lane == 0) red[wave] = w;
          __syncthreads();
          Peak best = red[0];
#pragma unroll
          for (int q = 1; q < NW; ++q) {
            const Peak o = red[q];
            if (peak_better(o.val, o.key, best.val, best.key)) best = o;
          }
          if (tid == 0) dst[round] = best.key >= 0 ? Peak{best.val * out_scale, best.key, best.key} : none;
          if (best.key >= 0 && (best.loc >> 4) == tid) taken |= 1u << (best.loc & 15);
        }
      }
    }
    if constexpr (!WIDE) {
      // This block's two largest first, its values taken in the order of their keys: ">=" is then the whole comparison
      // (the later of two equal values has the larger key).  e = 2 j + line.
      double bv1 = -INFINITY, bv2 = -INFINITY;
      int be1 = -1, be2 = -1;
      auto take = [&](double x, int e) {
        const bool g1 = x >= bv1, g2 = x >= bv2;
        bv2 = g1 ? bv1 : (g2 ? x : bv2);
        be2 = g1 ? be1 : (g2 ? e : be2);
        bv1 = g1 ? x : bv1;
        be1 = g1 ? e : be1;
      };
      if (d.transposed) {  // (uniform) key = l * S + s: element by element, line s0 before line s0 + 1
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (have0) take(v[j].x * cmp_scale, 2 * j);
          if (have1) take(v[j].y * cmp_scale, 2 * j + 1);
        }
      } else {  // key = s * N + l: line s0, then line s0 + 1
        if (have0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) take(v[j].x * cmp_scale, 2 * j);
        }
        if (have1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) take(v[j].y * cmp_scale, 2 * j + 1);
        }
      }
      auto key_of = [&](int e) {
        const int l = tid + NT * (e >> 1);
        const int key0 = d.transposed ? l * S + s0 : s0 * N + l;
        return e & 1 ? (d.transposed ? key0 + 1 : key0 + N) : key0;
      };
      offer(bv2, key_of(be2), be2 >= 0);
      offer(bv1, key_of(be1), be1 >= 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int l = tid + NT * j;
      const int key0 = d.transposed ? l * S + s0 : s0 * N + l;
      const int key1 = d.transposed ? key0 + 1 : key0 + N;
      if (want_abs) {
        const double a0v = fabs(v[j].x * scale), a1v = fabs(v[j].y * scale);
        const bool t0 = have0 & first_better(a0v, key0, ba.val, ba.key);
        ba.val = t0 ? a0v : ba.val;
        ba.key = t0 ? key0 : ba.key;
        const bool t1 = have1 & first_better(a1v, key1, ba.val, ba.key);
        ba.val = t1 ? a1v : ba.val;
        ba.key = t1 ? key1 : ba.key;
      }
    }
    s0 += 2;
    if (++it >= seq || s0 >= S) break;  // the same for every lane
    __syncthreads();                    // the last pass has read the line that the next one replaces
#if !(defined(MS_INV_ABLATE) && MS_INV_ABLATE == 2)  /* measurement build 2: the first block's lines again */
    load_lines();
#endif
    __syncthreads();
  }
  ba.loc = ba.key >= 0 ? 0 : -1;
  Peak b1{k1 >= 0 ? v1 * out_scale : -INFINITY, k1, k1}, b2{k2 >= 0 ? v2 * out_scale : -INFINITY, k2, k2};
  const Peak none{-INFINITY, -1, -1};
  if constexpr (ABS) {
    const Peak best = block_argmax_first<NW>(ba, red);
    if (tid == 0) partial_abs[(int64_t)blockIdx.y * abs_stride + blk0] = best;
    if (tid > 0 && tid < seq && blk0 + tid < nblk) partial_abs[(int64_t)blockIdx.y * abs_stride + blk0 + tid] = none;
    __syncthreads();
  }
  if constexpr (WIDE) return;  // the list entries were written where the values were ranked
  const Peak w1 = wave_best_peak(b1, peak_better);
  if (w1.loc >= 0 && w1.loc == b1.loc) b1 = b2;
  const Peak w2 = k > 1 ? wave_best_peak(b1, peak_better) : none;
  if ((tid & 63) == 0) {
    red[2 * (tid >> 6)] = w1;
    red[2 * (tid >> 6) + 1] = w2;
  }
  __syncthreads();
  if (tid == 0) {
    Peak t1 = none, t2 = none;
#pragma unroll
    for (int e = 0; e < 2 * NW; ++e) {
      const Peak o = red[e];
      if (peak_better(o.val, o.key, t1.val, t1.key)) {
        t2 = t1;
        t1 = o;
      } else if (peak_better(o.val, o.key, t2.val, t2.key)) {
        t2 = o;
      }
    }
    Peak* dst = partial + (int64_t)blockIdx.y * partial_stride + blk0 * k;
    dst[0] = t1;
    if (k > 1) dst[1] = t2;
  } else if (tid < seq && blk0 + tid < nblk) {  // the list entries of the other blocks this workgroup covered
    Peak* dst = partial + (int64_t)blockIdx.y * partial_stride + (blk0 + tid) * k;
    dst[0] = none;
    if (k > 1) dst[1] = none;
  }
}

// ---------------------------------------------------------------------------
// Strips whose sides are not powers of two: chirp-z (Bluestein) transforms built on
// the power-of-two engine, complex128 throughout, the whole (non-Hermitian) plane.
//   X[k] = c[k] * sum_j (x[j] c[j]) b[k - j],  c[j] = exp(-i pi j^2 / n),  b[m] = conj(c[|m|])
// The convolution runs as a length-M FFT (M = power of two >= 2n - 1) against the
// precomputed transform of b, stored in the engine's digit-reversed order and
// already divided by M.  The inverse transform conjugates c and b's transform.
struct GenPlanDev {
  int n, M;
  FftPlanDev planM;
  const double2* chirp;  // [n]
  const double2* bhat;   // [M], position order
};

// Transform nl lines of length n held in LDS (line f, element i at addr(f, i); the
// line has room for M elements), in place.  Block synchronised on entry and exit.
__device__ __forceinline__ void bluestein_lines(double2* buf, const LineAddr<double>& addr, int nl, const GenPlanDev& gp,
                                                const Twiddles<double>& tw, bool inverse, int tid, int nthr) {
  const int n = gp.n, M = gp.M;
  for (int e = tid; e < nl * M; e += nthr) {
    const int f = e / M, i = e - f * M;
    const int a = addr(f, i);
    if (i < n) {
      double2 c = gp.chirp[i];
      if (inverse) c.y = -c.y;
      buf[a] = cmul(buf[a], c);
    } else {
      buf[a] = mk<double>(0.0, 0.0);
    }
  }
  __syncthreads();
  fft_forward<double>(buf, addr, nl, gp.planM, tw, tid, nthr);
  for (int e = tid; e < nl * M; e += nthr) {
    const int f = e / M, pos = e - f * M;
    double2 h = gp.bhat[pos];
    if (inverse) h.y = -h.y;
    const int a = addr(f, pos);
    buf[a] = cmul(buf[a], h);
  }
  __syncthreads();
  fft_inverse<double>(buf, addr, nl, gp.planM, tw, tid, nthr);
  for (int e = tid; e < nl * n; e += nthr) {
    const int f = e / n, i = e - f * n;
    double2 c = gp.chirp[i];
    if (inverse) c.y = -c.y;
    const int a = addr(f, i);
    buf[a] = cmul(buf[a], c);
  }
  __syncthreads();
}

// The same with the chirp and filter values a lane needs held in registers (lane tid of an NT-wide
// workgroup only ever touches elements tid + NT j of a line) and no division in the element loops:
// for n <= kThreads * kBlueC and M <= kThreads * kBlueH.
#ifndef MS_LONG_INV_SEQ
#define MS_LONG_INV_SEQ 2
#endif
constexpr int kLongInvSeq = MS_LONG_INV_SEQ;  // blocks of lines one workgroup of the last pass may take in turn
constexpr int kMixedShortThreads = 256;  // radix-13 butterflies: 15 a column of 195, 240 for 16 columns
constexpr int kGenShortThreads = 512, kGenShortRadix = 8;  // as the power-of-two column pass: more waves, radix-8 plans
constexpr int kBlueC = 2, kBlueH = 4;
struct BluesteinRegs {
  double2 c[kBlueC];  // chirp[tid + NT j]
  double2 h[kBlueH];  // bhat[tid + NT j]
};
template <int NT>
__device__ __forceinline__ BluesteinRegs bluestein_load(const GenPlanDev& gp, int tid) {
  BluesteinRegs r;
#pragma unroll
  for (int j = 0; j < kBlueC; ++j) r.c[j] = gp.chirp[min(tid + NT * j, gp.n - 1)];
#pragma unroll
  for (int j = 0; j < kBlueH; ++j) r.h[j] = gp.bhat[min(tid + NT * j, gp.M - 1)];
  return r;
}
template <int NT, int MAXR>
__device__ __forceinline__ void bluestein_lines_r(double2* buf, const LineAddr<double>& addr, int nl, const GenPlanDev& gp,
                                                  const BluesteinRegs& r, const Twiddles<double>& tw, bool inverse, int tid) {
  const int n = gp.n, M = gp.M;
  const double sgn = inverse ? -1.0 : 1.0;
  for (int f = 0; f < nl; ++f) {
#pragma unroll
    for (int j = 0; j < kBlueH; ++j) {
      const int i = tid + NT * j;
      if (i < M) {
        const int a = addr(f, i);
        if (j < kBlueC && i < n)
          buf[a] = cmul(buf[a], mk<double>(r.c[j < kBlueC ? j : 0].x, sgn * r.c[j < kBlueC ? j : 0].y));
        else
          buf[a] = mk<double>(0.0, 0.0);
      }
    }
  }
  __syncthreads();
  fft_forward<double, MAXR>(buf, addr, nl, gp.planM, tw, tid, NT);
  for (int f = 0; f < nl; ++f) {
#pragma unroll
    for (int j = 0; j < kBlueH; ++j) {
      const int pos = tid + NT * j;
      if (pos < M) {
        const int a = addr(f, pos);
        buf[a] = cmul(buf[a], mk<double>(r.h[j].x, sgn * r.h[j].y));
      }
    }
  }
  __syncthreads();
  fft_inverse<double, MAXR>(buf, addr, nl, gp.planM, tw, tid, NT);
  for (int f = 0; f < nl; ++f) {
#pragma unroll
    for (int j = 0; j < kBlueC; ++j) {
      const int i = tid + NT * j;
      if (i < n) {
        const int a = addr(f, i);
        buf[a] = cmul(buf[a], mk<double>(r.c[j].x, sgn * r.c[j].y));
      }
    }
  }
  __syncthreads();
}

// The column pass for a short side that is not a power of two (the long side is one: the row passes stay
// as they are): the same unit structure, the length-S transforms done by chirp-z in LDS (complex128 only),
// columns as contiguous lines.  Rows come out in natural frequency order.
// grid (n_units, pairs_in_pass).
__global__ __launch_bounds__(kGenShortThreads) void k_fft_short_cps_gen(const PairDesc* desc, const int8_t* status, int pair0,
                                                                int S, int L, int want_transposed, GenPlanDev gpS,
                                                                FftPlanDev planL, const CpsUnit* __restrict__ units,
                                                                int lines, const double2* slabs, double2* spectrum,
                                                                double eps, double2* keep_r,
                                                                const uint16_t* __restrict__ pos2stored, int n_stored,
                                                                int exact) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  {
    const PairDesc& d = desc[p];
    if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  }
  const CpsUnit un = units[blockIdx.x];
  const int na = un.na, ncol = un.na + un.nb;
  const int tid = threadIdx.x;
  const int Lp = spectrum_pitch<double>(n_stored);
  double2* Z = spectrum + (int64_t)blockIdx.y * S * Lp;
  const int64_t sp = slab_pitch(S);
  const double2* Zs = slabs + (int64_t)blockIdx.y * slab_pair_elems(S, L);
  const LineAddr<double> addr{LineAddr<double>::padded(gpS.M) + 1};
  const Twiddles<double> tw = load_twiddles<double>(buf + lines * addr.fstride, gpS.planM, tid, kGenShortThreads);
  const int lncol = 31 - __builtin_clz(ncol), lna = 31 - __builtin_clz(na);
  const int mp = planL.mpos[un.a0 + (tid & (na - 1))];  // column c of a lane's cells is the same on every trip
  const int col0 = pos2stored[un.a0];
  const BluesteinRegs br = bluestein_load<kGenShortThreads>(gpS, tid);
  for (int e = tid; e < S * ncol; e += kGenShortThreads) {
    const int q = e >> lncol, c = e & (ncol - 1);
    const int pos = c < na ? un.a0 + c : un.b0 + (c - na);
    buf[addr(c, q)] = Zs[(pos / kSlab) * sp + q * kSlab + pos % kSlab];
  }
  __syncthreads();
  bluestein_lines_r<kGenShortThreads, kGenShortRadix>(buf, addr, ncol, gpS, br, tw, false, tid);
  const bool partner_in_a = mp >= un.a0 && mp < un.a0 + na;
  const int pc = partner_in_a ? (mp - un.a0) : (na + mp - un.b0);
  for (int e = tid; e < S * na; e += kGenShortThreads) {
    const int q = e >> lna, c = e & (na - 1);
    const int q2 = q == 0 ? 0 : S - q;
    if (partner_in_a && (q2 * 64 + pc) < (q * 64 + c)) continue;  // the partner cell's owner writes both
    const double2 z = buf[addr(c, q)];
    const double2 zm = buf[addr(pc, q2)];
    double ax = 0.5 * (z.x + zm.x), ay = 0.5 * (z.y - zm.y);
    double bx = 0.5 * (z.y + zm.y), by = -0.5 * (z.x - zm.x);
    if (exact && (q2 * 64 + pc) == (q * 64 + c)) exact_real_bin(ax, bx);  // its own partner
    double fx = ax * bx + ay * by;  // A * conj(B)
    double fy = ay * bx - ax * by;
    const double inv = 1.0 / (cabs_safe<double>(fx, fy) + eps);
    fx *= inv;
    fy *= inv;
    buf[addr(c, q)] = mk<double>(fx, fy);
    if (partner_in_a && (q2 * 64 + pc) != (q * 64 + c)) buf[addr(pc, q2)] = mk<double>(fx, -fy);
  }
  __syncthreads();
  if (keep_r != nullptr) {
    // the subpixel stage's copy: rows in ascending signed frequency from row Sr / 2 - S / 2 on; an odd S
    // leaves the first of its Sr = S + 1 rows empty
    double2* Rs = keep_r + (int64_t)blockIdx.y * slab_pair_elems(S, L);
    const int Sr = S + (S & 1);
    for (int e = tid; e < S * na; e += kGenShortThreads) {
      const int q = e >> lna, c = e & (na - 1);
      const int pos = un.a0 + c;
      const int f = q < (S + 1) / 2 ? q : q - S;
      Rs[(pos / kSlab) * sp + (f + Sr / 2) * kSlab + pos % kSlab] = buf[addr(c, q)];
    }
    if ((S & 1) && tid < na) {
      const int pos = un.a0 + tid;
      Rs[(pos / kSlab) * sp + pos % kSlab] = mk<double>(0.0, 0.0);
    }
    __syncthreads();
  }
  bluestein_lines_r<kGenShortThreads, kGenShortRadix>(buf, addr, na, gpS, br, tw, true, tid);
  for (int e = tid; e < S * na; e += kGenShortThreads) {
    const int q = e >> lna, c = e & (na - 1);
    Z[(int64_t)q * Lp + col0 + c] = buf[addr(c, q)];
  }
}

// z = a + i b per line, length-L transform, natural order out.  grid (S, pairs_in_pass).
__global__ __launch_bounds__(kThreads) void k_gen_rows_fwd(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                           int L, int want_transposed, GenPlanDev gp, double2* Zf,
                                                           const double* __restrict__ win_s,
                                                           const double* __restrict__ win_l) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  const int tid = threadIdx.x;
  const LineAddr<double> addr{LineAddr<double>::padded(gp.M) + 1};
  const Twiddles<double> tw = load_twiddles<double>(buf + addr.fstride, gp.planM, tid, kThreads);
  const int s = blockIdx.x;
  const double ws = win_s != nullptr ? win_s[s] : 1.0;  // optional separable window (none in the reference)
  for (int i = tid; i < L; i += kThreads) {
    const double w = win_l != nullptr ? ws * win_l[i] : 1.0;
    buf[addr(0, i)] = mk<double>((double)d.a[(int64_t)s * d.pitch + i] * w, (double)d.b[(int64_t)s * d.pitch + i] * w);
  }
  __syncthreads();
  bluestein_lines(buf, addr, 1, gp, tw, false, tid, kThreads);
  double2* out = Zf + ((int64_t)blockIdx.y * S + s) * L;
  for (int i = tid; i < L; i += kThreads) out[i] = buf[addr(0, i)];
}

// Length-S transform down C adjacent columns of the [S][L] plane, in place.
// grid (ceil(L / C), pairs_in_pass).
__global__ __launch_bounds__(kThreads) void k_gen_cols(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                       int L, int want_transposed, GenPlanDev gp, int C, int inverse,
                                                       double2* Z) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  {
    const PairDesc& d = desc[p];
    if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  }
  const int tid = threadIdx.x;
  const LineAddr<double> addr{LineAddr<double>::padded(gp.M) + 1};
  const Twiddles<double> tw = load_twiddles<double>(buf + C * addr.fstride, gp.planM, tid, kThreads);
  const int c0 = blockIdx.x * C;
  const int nc = min(C, L - c0);
  double2* plane = Z + (int64_t)blockIdx.y * S * L;
  for (int e = tid; e < S * nc; e += kThreads) {
    const int q = e / nc, c = e - q * nc;
    buf[addr(c, q)] = plane[(int64_t)q * L + c0 + c];
  }
  __syncthreads();
  bluestein_lines(buf, addr, nc, gp, tw, inverse != 0, tid, kThreads);
  for (int e = tid; e < S * nc; e += kThreads) {
    const int q = e / nc, c = e - q * nc;
    plane[(int64_t)q * L + c0 + c] = buf[addr(c, q)];
  }
}

// Normalised cross-power spectrum of the two real strips packed in Zf = F(a + i b).
// grid (blocks, pairs_in_pass).
__global__ __launch_bounds__(kThreads) void k_gen_cps(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                      int L, int want_transposed, const double2* __restrict__ Zf,
                                                      double2* __restrict__ R, double eps, int exact) {
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  {
    const PairDesc& d = desc[p];
    if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  }
  const double2* zf = Zf + (int64_t)blockIdx.y * S * L;
  double2* r = R + (int64_t)blockIdx.y * S * L;
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < S * L; e += gridDim.x * kThreads) {
    const int q = e / L, c = e - q * L;
    const double2 z = zf[e];
    const double2 zm = zf[(int64_t)((S - q) % S) * L + (L - c) % L];
    double ax = 0.5 * (z.x + zm.x), ay = 0.5 * (z.y - zm.y);
    double bx = 0.5 * (z.y + zm.y), by = -0.5 * (z.x - zm.x);
    if (exact && (S - q) % S == q && (L - c) % L == c) exact_real_bin(ax, bx);  // its own partner
    double fx = ax * bx + ay * by;  // A * conj(B)
    double fy = ay * bx - ax * by;
    const double inv = 1.0 / (cabs_safe<double>(fx, fy) + eps);
    r[e] = mk<double>(fx * inv, fy * inv);
  }
}

// Copy of the normalised spectrum for the subpixel stage, in the layout its kernels read: the columns of
// frequency 0 .. L/2 in slabs of 8, rows in ascending signed frequency (an odd S gets an empty first row,
// which also stands in for the Nyquist row it does not have).  grid (blocks, pairs_in_pass).
__global__ __launch_bounds__(kThreads) void k_gen_keep_r(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                         int L, int want_transposed, const double2* __restrict__ R,
                                                         int Sr, int n_cols8, int64_t pair_stride,
                                                         double2* __restrict__ out) {
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  {
    const PairDesc& d = desc[p];
    if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  }
  const double2* r = R + (int64_t)blockIdx.y * S * L;
  double2* o = out + (int64_t)blockIdx.y * pair_stride;
  const int64_t sp = slab_pitch(Sr);
  const int n_cols = L / 2 + 1;
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < Sr * n_cols8; e += gridDim.x * kThreads) {
    const int kk = e / n_cols8, c = e - kk * n_cols8;
    const int f = kk - Sr / 2;                  // signed frequency of the row
    const int q = f < 0 ? f + S : f;            // its index in the transform
    const bool have = c < n_cols && -f <= S / 2 && f < (S + 1) / 2;
    o[(c / kSlab) * sp + kk * kSlab + c % kSlab] = have ? r[(int64_t)q * L + c] : mk<double>(0.0, 0.0);
  }
}

// Inverse length-L transform of one line, scale, the line's k largest values.
// grid (S, pairs_in_pass).
__global__ __launch_bounds__(kThreads) void k_gen_rows_inv(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                           int L, int want_transposed, GenPlanDev gp,
                                                           const double2* __restrict__ R, int k, Peak* partial,
                                                           int partial_stride, double* pcm_out, Peak* partial_abs,
                                                           int abs_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double2* buf = reinterpret_cast<double2*>(smem_raw);
  __shared__ Peak red[kThreads / 64];
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  const int tid = threadIdx.x;
  const LineAddr<double> addr{LineAddr<double>::padded(gp.M) + 1};
  const Twiddles<double> tw = load_twiddles<double>(buf + addr.fstride, gp.planM, tid, kThreads);
  const int s = blockIdx.x;
  const double2* in = R + ((int64_t)blockIdx.y * S + s) * L;
  for (int i = tid; i < L; i += kThreads) buf[addr(0, i)] = in[i];
  __syncthreads();
  bluestein_lines(buf, addr, 1, gp, tw, true, tid, kThreads);
  const double scale = 1.0 / ((double)S * (double)L);
  if (pcm_out != nullptr) {
    double* out = pcm_out + (int64_t)blockIdx.y * S * L;
    for (int l = tid; l < L; l += kThreads)
      out[d.transposed ? (int64_t)l * S + s : (int64_t)s * L + l] = buf[addr(0, l)].x * scale;
  }
  if (partial_abs != nullptr) {  // the line's largest magnitude, first in row-major order (subpixel stage)
    Peak ba{-INFINITY, -1, -1};
    for (int l = tid; l < L; l += kThreads) {
      const double av = fabs(buf[addr(0, l)].x * scale);
      const int key = d.transposed ? l * S + s : s * L + l;
      if (first_better(av, key, ba.val, ba.key)) ba = Peak{av, key, 0};
    }
    const Peak best = block_argmax_first(ba, red);
    if (tid == 0) partial_abs[(int64_t)blockIdx.y * abs_stride + blockIdx.x] = best;
    __syncthreads();
  }
  // k rounds of block argmax; a taken element is overwritten with -inf
  for (int round = 0; round < k; ++round) {
    Peak mine{-INFINITY, -1, -1};
    for (int l = tid; l < L; l += kThreads) {
      const int a0 = addr(0, l);
      const double raw = buf[a0].x;
      const int key = d.transposed ? l * S + s : s * L + l;
      if (raw != -INFINITY && peak_better(raw * scale, key, mine.val, mine.key)) mine = Peak{raw * scale, key, a0};
    }
    const Peak best = block_argmax(mine, red);
    if (tid == 0) {
      partial[(int64_t)blockIdx.y * partial_stride + blockIdx.x * k + round] = best;
      if (best.loc >= 0) buf[best.loc].x = -INFINITY;
    }
    __syncthreads();
  }
}

// Merge the per-block peaks of one pair and emit the candidate table.
// grid (n_pairs_in_pass), block kThreads.
__global__ __launch_bounds__(kThreads) void k_peaks_candidates(const PairDesc* desc, const int8_t* status,
                                                               int pair0, Peak* partial, int partial_stride,
                                                               int nblk_direct, int nblk_transposed, int k,
                                                               int n_peaks_used, Peak* peaks /*[P][kMaxK]*/,
                                                               Cand* cand /*[P][16 n_peaks_used + 1]*/,
                                                               unsigned long long* acc /*[P][16 n_peaks_used + 1][5]*/) {
  __shared__ Peak red[kThreads / 64];
  __shared__ Peak top[kMaxK];
  const int p = pair0 + blockIdx.x;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  const int tid = threadIdx.x;
  Peak* mine_list = partial + (int64_t)blockIdx.x * partial_stride;
  const int n_entries = (d.transposed ? nblk_transposed : nblk_direct) * k;  // what k_fft_long_inv wrote
  for (int round = 0; round < k; ++round) {
    Peak mine{-INFINITY, -1, -1};
    for (int e = tid; e < n_entries; e += kThreads) {
      const Peak c = mine_list[e];
      if (c.key >= 0 && c.loc != -2 && peak_better(c.val, c.key, mine.val, mine.key)) mine = Peak{c.val, c.key, e};
    }
    Peak best = block_argmax(mine, red);
    if (tid == 0) {
      top[round] = best;
      if (best.loc >= 0) mine_list[best.loc].loc = -2;  // taken
      peaks[(int64_t)p * kMaxK + round] = best;
    }
    __syncthreads();
  }
  if (cand == nullptr) return;
  const int n_cand = 16 * n_peaks_used, slots = n_cand + 1;
  static_assert(kNumCandWide < kThreads, "one thread per candidate and one for the subpixel slot");
  if (tid == n_cand) {  // the subpixel slot stays unused until k_ups_finish fills it
    cand[(int64_t)p * slots + n_cand] = Cand{0, 0, 1, -1};
    for (int i = 0; i < 5; ++i) acc[((int64_t)p * slots + n_cand) * 5 + i] = 0ull;
  }
  __shared__ Cand sc[kNumCandWide];
  if (tid < n_cand) {
    const int pk = tid >> 4, t = tid & 15;
    Cand c{0, 0, 1, -1};
    if (pk < n_peaks_used && pk < k && top[pk].key >= 0) {
      const int py = top[pk].key / d.img_w, px = top[pk].key - py * d.img_w;
      const int ymag = (t & 8) ? d.img_h - py : py;
      const int xmag = (t & 4) ? d.img_w - px : px;
      c.y = (t & 2) ? ymag : -ymag;
      c.x = (t & 1) ? xmag : -xmag;
      const int ay = c.y < 0 ? -c.y : c.y, ax = c.x < 0 ? -c.x : c.x;
      if (ay >= d.img_h || ax >= d.img_w) {
        c.flag = 1;
      } else if ((long long)(d.img_h - ay) * (d.img_w - ax) < 25) {
        c.flag = 2;
      } else {
        c.flag = 0;
      }
    }
    sc[tid] = c;
    for (int i = 0; i < 5; ++i) acc[((int64_t)p * slots + tid) * 5 + i] = 0ull;
  }
  __syncthreads();
  if (tid < n_cand) {
    Cand c = sc[tid];
    if (c.flag == 0) {
      for (int e = 0; e < tid; ++e) {
        const Cand o = sc[e];
        if (o.flag == 0 && o.y == c.y && o.x == c.x) {
          c.flag = 3;
          c.dup_of = e;
          break;
        }
      }
    }
    cand[(int64_t)p * slots + tid] = c;
  }
}

// ---------------------------------------------------------------------------
// Subpixel refinement by upsampled DFT (the matrix-multiply refinement of
// Guizar-Sicairos, Thurman and Fienup, as skimage.registration.phase_cross_correlation
// runs it for registration/_translation_computation.py:1233-1272): the inverse DFT of
// the normalised cross-power spectrum R is re-evaluated on an n_up x n_up grid of
// spacing 1/u around the coarse peak,
//   CC(ds, dl) = sum_{a,b} R[a,b] exp(2 pi i (ds ks(a) / S + dl kl(b) / L)),
// ks, kl the signed frequencies (Nyquist counted negative, as numpy.fft.fftfreq does).
// Only one column b of every conjugate pair {b, -b} is stored.  With
//   X_b(ds) = sum_a R[a,b] exp(2 pi i ds ks(a) / S)          (first stage, k_ups_short)
// the pair contributes 2 Re(X_b e_b) + c_b conj(e_b), e_b = exp(2 pi i dl kl(b) / L),
// where c_b = -2i sin(pi ds) conj(R[S/2, b]) accounts for the Nyquist row, whose signed
// frequency is not negated by conjugation; self-conjugate columns contribute X_b e_b
// (second stage, k_ups_long).
//
// Every term of CC is exp(i (ws ds + wl dl)) with |ws|, |wl| <= pi, and the grid spans only
// h = (n_up - 1) / (2 u) ~ 0.75 pixels either side of its centre: along each axis CC is an entire function of
// exponential type pi h ~ 2.4 on [-1, 1], which the polynomial through kUpsNodes = 32 Chebyshev nodes reproduces
// to rounding error (truncation 2 (pi h / 2)^32 / 32! ~ 1e-33 of the largest possible |CC|, Lebesgue constant 3.2;
// measured against the direct sums: 2e-15 of the peak).  The two stages therefore evaluate CC at the 32 x 32 node
// displacements only (a fifth of the first stage's products, a twentieth of the second's) and k_ups_interp
// carries the node values to the n_up x n_up grid with the barycentric interpolation matrix P (n_up x 32, built
// on the host in extended precision): CC_grid = P CC_nodes P^T.
// Coarse shift of the subpixel stage: the pair's largest |PCM| value (first in row-major
// order on ties), wrapped to signed about the midpoint.  grid (pairs_in_pass).
__global__ __launch_bounds__(kThreads) void k_ups_coarse(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                         int L, int want_transposed, const Peak* partial_abs,
                                                         int abs_stride, int nblk, UpsState* ups) {
  __shared__ Peak red[kThreads / 64];
  const int p = pair0 + blockIdx.x;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  Peak mine{-INFINITY, -1, -1};
  for (int e = threadIdx.x; e < nblk; e += kThreads) {
    const Peak c = partial_abs[(int64_t)blockIdx.x * abs_stride + e];
    if (first_better(c.val, c.key, mine.val, mine.key)) mine = c;
  }
  const Peak best = block_argmax_first(mine, red);
  if (threadIdx.x == 0) {
    UpsState u{0, 0, 0, 0, 0.0, 0.0};
    if (best.key >= 0) {
      int sy = best.key / d.img_w, sx = best.key - sy * d.img_w;
      if (sy > d.img_h / 2) sy -= d.img_h;
      if (sx > d.img_w / 2) sx -= d.img_w;
      u.shift_s = d.transposed ? sx : sy;
      u.shift_l = d.transposed ? sy : sx;
      u.valid = 1;
    }
    ups[p] = u;
  }
}

// One rotor step w * s with fused multiply-adds (four instead of six float64 operations: the rotors
// share the float64 units with the matrix instructions they feed).
__device__ __forceinline__ double2 rotate(double2 w, double2 s) {
  return mk<double>(__builtin_fma(w.x, s.x, -(w.y * s.y)), __builtin_fma(w.x, s.y, w.y * s.x));
}

struct KappaEntry {
  uint16_t stored;  // index into the stored-position list of the column with |kl| = kappa
  int8_t sign;      // +1: its signed frequency is +kappa, -1: -kappa
  int8_t self;      // 1: the column is its own conjugate partner (kappa = 0 or L / 2)
};

// First stage on the matrix cores.  For one slab (8 columns) X[u][c] = sum_kk R[kk][c] E[u][kk],
// E[u][kk] = exp(i theta_u (kk - S/2)), u the kUpsNodes node displacements, is a real product C = A B with
//   A[u][2 kk + p]       = p ? Im E : Re E                       (generated: one rotor per lane)
//   B[2 kk + p][2 c + o] = (p == o) ? Re R : (p ? -Im R : Im R)  (one double per lane from LDS)
//   C[u][2 c + o]        = o ? Im X : Re X
// run as v_mfma_f64_16x16x4_f64 steps: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15];
// result register g of lane l is C[(l >> 4) + 4 g][l & 15].  A step covers two kk.
// grid (ceil(n_slabs / NS), pairs_in_pass), four waves.  A workgroup holds NS slabs in LDS.
typedef double d4 __attribute__((ext_vector_type(4)));
static_assert(kUpsNodes == 32, "two 16-row tiles of node displacements");

template <typename T, int NS>
__global__ __launch_bounds__(kThreads) void k_ups_short(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                        int L, int want_transposed, int Sr, int64_t pair_stride,
                                                        const uint16_t* __restrict__ slab_pos, int n_slabs,
                                                        const uint16_t* __restrict__ pos2stored,
                                                        const uint16_t* __restrict__ stored2kappa, int n_stored,
                                                        const cx<T>* __restrict__ rslabs,
                                                        const UpsState* __restrict__ ups, const double* __restrict__ node,
                                                        double2* __restrict__ T1, double2* __restrict__ Nq) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* rows = reinterpret_cast<double*>(smem_raw);  // [NS][S][2 * kSlab], rows in ascending signed frequency
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  {
    const PairDesc& d = desc[p];
    if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  }
  const UpsState st = ups[p];
  if (!st.valid) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Sr rows per slab, row kk holding signed frequency kk - Sr / 2 (Sr = S, or S + 1 with an empty first row
  // when S is odd)
  const cx<T>* Rs = rslabs + (int64_t)blockIdx.y * pair_stride;
  const int64_t sp = slab_pitch(Sr);
  const int slab0 = blockIdx.x * NS;
  const int slab_elems = Sr * 2 * kSlab;  // doubles per slab in LDS
  if (tid < NS * kSlab && slab0 + tid / kSlab < n_slabs) {
    // the Nyquist row (signed frequency -S/2, the first row) of every column
    const int pos = slab_pos[slab0 + tid / kSlab] + tid % kSlab;
    const cx<T> v = Rs[(pos / kSlab) * sp + pos % kSlab];
    const int col = pos2stored[pos];  // 0xffff: a column of the group that is not stored
    if (col != 0xffff) Nq[(int64_t)blockIdx.y * n_stored + col] = mk<double>((double)v.x, (double)v.y);
  }
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const bool have = slab0 + ns < n_slabs;  // a missing slab is computed as zeros and not stored
    const cx<T>* slab = Rs + (slab_pos[min(slab0 + ns, n_slabs - 1)] / kSlab) * sp;
    double* dst = rows + ns * slab_elems;
    constexpr int U = 8;  // loads in flight per lane: one at a time would leave the workgroup waiting on memory
    for (int e0 = tid; e0 < Sr * kSlab; e0 += kThreads * U) {
      cx<T> v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = slab[min(e0 + u * kThreads, Sr * kSlab - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        MS_PIN(v[u].x);
        MS_PIN(v[u].y);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        if (e < Sr * kSlab) {
          dst[2 * e] = have ? (double)v[u].x : 0.0;
          dst[2 * e + 1] = have ? (double)v[u].y : 0.0;
        }
      }
    }
  }
  __syncthreads();
  // results go to T1[|frequency| of the column][node]: what the second stage reads is then contiguous, and
  // the 16 displacements a lane group holds for one column are adjacent here
  double2* t1 = T1 + (int64_t)blockIdx.y * n_stored * kUpsNodes;
  // lane roles inside a step
  const int i = lane & 15, k = lane >> 4;  // A: row i, k index; B: k index, column i
  const int part = k & 1, kk_off = k >> 1;
  const int comp = (part != (i & 1)) ? 1 : 0;            // which component of R this lane feeds to B
  const int bcol = 2 * (i >> 1) + comp;
  // wave w: displacement tile w & 1, half w >> 1 of the kk range, every slab (one rotor update then serves NS
  // products); the two halves are added through LDS at the end
  const int ut = wave & 1, half = wave >> 1;
  const int all_steps = Sr / 2;  // (Sr is even)
  const int first_steps = (all_steps + 1) / 2;
  const int step0 = half ? first_steps : 0, nsteps = half ? all_steps - first_steps : first_steps;
  const long long smask = (part == 1 && (i & 1) == 0) ? (long long)0x8000000000000000ull : 0ll;  // -Im R
  d4 acc[NS];
  double2 w, step2;
  {
    const double delta = (double)st.shift_s + node[ut * 16 + i];
    const double theta = 2.0 * 3.14159265358979323846 * delta / (double)S;
    double sn, cs;
    sincos(theta * (double)(2 * step0 + kk_off - Sr / 2), &sn, &cs);
    w = mk<double>(cs, sn);
    sincos(2.0 * theta, &sn, &cs);
    step2 = mk<double>(cs, sn);
  }
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) acc[ns] = d4{0.0, 0.0, 0.0, 0.0};
  if (nsteps > 0) {
    const double* brow = rows + (2 * step0 + kk_off) * (2 * kSlab) + bcol;
    double nb[NS];  // the next step's operands are read from LDS while this step's products run
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) nb[ns] = brow[ns * slab_elems];
    for (int sidx = 0; sidx < nsteps; ++sidx) {
      double bv[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bv[ns] = __longlong_as_double(__double_as_longlong(nb[ns]) ^ smask);
      const int nrow = 2 * min(sidx + 1, nsteps - 1);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) nb[ns] = brow[ns * slab_elems + nrow * (2 * kSlab)];
      const double av = part ? w.y : w.x;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[ns], acc[ns], 0, 0, 0);
      w = rotate(w, step2);
    }
  }
  __syncthreads();  // every wave has read its last row: the slabs make room for the second half's sums
  double* red = rows;  // [tile][slab][register][lane]
  if (half == 1) {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int g = 0; g < 4; ++g) red[((ut * NS + ns) * 4 + g) * 64 + lane] = acc[ns][g];
  }
  __syncthreads();
  if (half == 1) return;
  const int c = i >> 1, o = i & 1;
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    if (slab0 + ns >= n_slabs) continue;
    const int col = pos2stored[slab_pos[slab0 + ns] + c];
    if (col == 0xffff) continue;  // not a stored column
    const int kap = stored2kappa[col];
    if (kap == 0xffff) continue;  // the second stored column of a conjugate pair: unused
#pragma unroll
    for (int g = 0; g < 4; ++g)
      reinterpret_cast<double*>(t1 + (int64_t)kap * kUpsNodes + ut * 16 + k + 4 * g)[o] =
          acc[ns][g] + red[((ut * NS + ns) * 4 + g) * 64 + lane];
  }
}

// The first stage with the rows of E in registers.  Every slab of a pair is multiplied by the same E, so a workgroup
// that takes G slabs in turn generates its part of E once (one rotor chain per lane) and the products then issue
// back to back: in k_ups_short a rotor update (four float64 vector instructions, which do not overlap the float64
// matrix instructions) costs as much as the product it feeds.  Eight waves: wave w takes the displacement tile w & 1
// and quarter w >> 1 of the kk range, QS steps of two kk at most (a lane's QS values of E are QS registers pairs);
// the quarters are added through LDS.  The next slab's values are requested before the current slab's products and
// written to LDS while the sums of the current one are added and stored (two barriers per slab).
// grid (ceil(n_slabs / G), pairs_in_pass), 512 lanes; LDS: one slab, the sums of the eight waves, the column tables.
constexpr int kUpsShortThreads = 512;
constexpr int kUpsGroupMax = 64;  // slabs a workgroup takes in turn at most
template <typename T, int QS>
__global__ __launch_bounds__(kUpsShortThreads) void k_ups_short_reg(const PairDesc* desc, const int8_t* status, int pair0,
                                                                    int S, int L, int want_transposed, int Sr,
                                                                    int64_t pair_stride,
                                                                    const uint16_t* __restrict__ slab_pos, int n_slabs,
                                                                    const uint16_t* __restrict__ pos2stored,
                                                                    const uint16_t* __restrict__ stored2kappa,
                                                                    int n_stored, const cx<T>* __restrict__ rslabs,
                                                                    const UpsState* __restrict__ ups,
                                                                    const double* __restrict__ node,
                                                                    double2* __restrict__ T1, double2* __restrict__ Nq,
                                                                    int G) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* rows = reinterpret_cast<double*>(smem_raw);  // [Sr][2 * kSlab], rows in ascending signed frequency
  double* red = rows + (size_t)Sr * 2 * kSlab;         // [quarter][tile][register][lane]
  // of the workgroup's slabs: first storage position, and per column the stored column (0xffff: not stored) and its
  // |frequency| (0xffff: not stored, or the second column of a conjugate pair) - looked up once, not per slab
  uint16_t* s_pos = reinterpret_cast<uint16_t*>(red + 8 * 4 * 64);  // [kUpsGroupMax]
  uint16_t* s_col = s_pos + kUpsGroupMax;                           // [kUpsGroupMax][kSlab]
  uint16_t* s_kap = s_col + kUpsGroupMax * kSlab;                   // [kUpsGroupMax][kSlab]
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  {
    const PairDesc& d = desc[p];
    if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  }
  const UpsState st = ups[p];
  if (!st.valid) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const cx<T>* Rs = rslabs + (int64_t)blockIdx.y * pair_stride;
  const int64_t sp = slab_pitch(Sr);
  const int slab0 = blockIdx.x * G, slab1 = min(slab0 + G, n_slabs);
  double2* t1 = T1 + (int64_t)blockIdx.y * n_stored * kUpsNodes;
  // lane roles inside a step (see k_ups_short)
  const int i = lane & 15, k = lane >> 4;
  const int part = k & 1, kk_off = k >> 1;
  const int comp = (part != (i & 1)) ? 1 : 0;
  const int bcol = 2 * (i >> 1) + comp;
  const long long smask = (part == 1 && (i & 1) == 0) ? (long long)0x8000000000000000ull : 0ll;  // -Im R
  const int ut = wave & 1, quarter = wave >> 1;
  const int all_steps = Sr / 2;  // (Sr is even)
  const int per = (all_steps + 3) / 4;  // <= QS, checked by the host
  const int step0 = min(all_steps, quarter * per), nsteps = min(all_steps, step0 + per) - step0;
  constexpr int U = 8 * QS * kSlab / kUpsShortThreads;  // loads per lane and slab (Sr <= 8 QS)
  cx<T> pre[U];
  for (int e = tid; e < (slab1 - slab0) * kSlab; e += kUpsShortThreads) {
    const int pos = slab_pos[slab0 + e / kSlab];
    const int col = pos2stored[pos + e % kSlab];
    if (e % kSlab == 0) s_pos[e / kSlab] = (uint16_t)pos;
    s_col[e] = (uint16_t)col;
    s_kap[e] = col == 0xffff ? (uint16_t)0xffff : stored2kappa[col];
  }
  auto fetch = [&](int sl, int pos) {
#if defined(MS_UPS_ABLATE) && MS_UPS_ABLATE == 1  /* measurement build: one slab of every pair is read over and over */
    pos = 0;
#endif
    const cx<T>* slab = Rs + (pos / kSlab) * sp;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const cx<T>* at = slab + min(tid + u * kUpsShortThreads, Sr * kSlab - 1);
      if constexpr (std::is_same<T, double>::value) pre[u] = stream_load<128>(at);
      else pre[u] = *at;
    }
  };
  fetch(slab0, slab_pos[slab0]);
  double av[QS];
  {
    const double delta = (double)st.shift_s + node[ut * 16 + i];
    const double theta = 2.0 * 3.14159265358979323846 * delta / (double)S;
    double sn, cs;
    sincos(theta * (double)(2 * step0 + kk_off - Sr / 2), &sn, &cs);
    double2 w = mk<double>(cs, sn);
    sincos(2.0 * theta, &sn, &cs);
    const double2 step2 = mk<double>(cs, sn);
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      av[q] = part ? w.y : w.x;
      w = rotate(w, step2);
    }
  }
  const double* brow = rows + (2 * step0 + kk_off) * (2 * kSlab) + bcol;
  const int c = i >> 1, o = i & 1;
  auto spread = [&](int sl) {  // the fetched slab into LDS, its Nyquist row (signed frequency -S/2, the first row) to Nq
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + u * kUpsShortThreads;
      if (e < Sr * kSlab) {
        rows[2 * e] = (double)pre[u].x;
        rows[2 * e + 1] = (double)pre[u].y;
      }
    }
    if (tid < kSlab) {
      const int col = s_col[(sl - slab0) * kSlab + tid];
      if (col != 0xffff) Nq[(int64_t)blockIdx.y * n_stored + col] = mk<double>((double)pre[0].x, (double)pre[0].y);
    }
  };
  __syncthreads();  // the tables
  spread(slab0);
  __syncthreads();
  for (int sl = slab0; sl < slab1; ++sl) {
    if (sl + 1 < slab1) fetch(sl + 1, s_pos[sl + 1 - slab0]);  // in flight during the products
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#if defined(MS_UPS_ABLATE) && MS_UPS_ABLATE == 2  /* measurement build: no products */
    if (nsteps > 1000000)
#endif
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (q < nsteps) {  // (the same for every lane of the wave)
        const double b = __longlong_as_double(__double_as_longlong(brow[2 * q * (2 * kSlab)]) ^ smask);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], b, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((quarter * 2 + ut) * 4 + g) * 64 + lane] = acc[g];
    __syncthreads();  // every wave has read its last row of the slab and written its sums
    // (the next slab first: its values were requested a whole round of products ago, and waiting for them after
    // the stores below would wait for those stores as well)
    if (sl + 1 < slab1) spread(sl + 1);
    {
      // wave (tile, quarter) adds the four quarters of result register g = quarter of its tile and stores it
      const int g = quarter;
      const int kap = s_kap[(sl - slab0) * kSlab + c];
      double v = red[((0 * 2 + ut) * 4 + g) * 64 + lane];
#pragma unroll
      for (int qq = 1; qq < 4; ++qq) v += red[((qq * 2 + ut) * 4 + g) * 64 + lane];
      if (kap != 0xffff) reinterpret_cast<double*>(t1 + (int64_t)kap * kUpsNodes + ut * 16 + k + 4 * g)[o] = v;
    }
    __syncthreads();  // the next slab is in place, the sums have been consumed
  }
}

// Second stage on the matrix cores.  With Y[us][kappa] the first-stage value of the stored column of
// |frequency| kappa (conjugated where that column's frequency is negative, doubled unless it is its own
// partner), Re CC[us][ul] = sum_kappa Re(Y w), w = exp(i theta_ul kappa), is a real product C = A B with
//   A[us][2 kappa + p] = p ? -Im Y : Re Y        (staged in LDS, 32 kappa at a time)
//   B[2 kappa + p][ul] = p ? Im w : Re w         (generated: one rotor per lane)
// run as v_mfma_f64_16x16x4_f64 steps of two kappa (lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15];
// result register g of lane l is C[(l >> 4) + 4 g][l & 15]).  The imaginary part exists only for the two
// self-conjugate columns.
// grid (kUpsLongParts, pairs_in_pass), five waves: a workgroup sums one part of the kappa range (a pass of 128
// pairs would otherwise be 128 workgroups, each a chain of 500 dependent steps); waves 0-3 take the 2 x 2 result
// tiles of the 32 x 32 node displacements (wave w: column tile w & 1, row tile w >> 1) and write the part's sums to
// ccn_part[pair][part][us][ul].  Wave 4 forms the part's Nyquist-row sums for every long-axis node ul,
//   g(ul) = g1 + conj(g2),  g1 = sum over stored columns of negative frequency of V w,  g2 = the same for
//   positive ones, V = the column's Nyquist-row value (conjugated where the stored frequency is negative),
// into g_part[pair][part][ul] (lanes: node ul = lane & 31, half lane >> 5 of each round's 32 kappa).
// k_ups_interp adds the parts in a fixed order.
constexpr int kUpsLongParts = 8;
constexpr int kUpsLongThreads = kThreads + 64;
__global__ __launch_bounds__(kUpsLongThreads) void k_ups_long(const PairDesc* desc, const int8_t* status, int pair0, int S,
                                                              int L, int want_transposed,
                                                              const KappaEntry* __restrict__ kappa, int n_stored,
                                                              const double2* __restrict__ T1,
                                                              const double2* __restrict__ Nq,
                                                              const UpsState* __restrict__ ups,
                                                              const double* __restrict__ node,
                                                              double2* __restrict__ ccn_part,
                                                              double2* __restrict__ g_part) {
  constexpr int kChunk = 32;  // kappa per staging round
  static_assert(kUpsNodes == 32 && kThreads == 256, "tile assignment below");
  __shared__ double sA[kChunk][kUpsNodes][2];  // (Re Y, -Im Y)
  __shared__ double2 sSelf[2][kUpsNodes];      // Y of the self-conjugate columns kappa = 0 and kappa = L / 2
  __shared__ double2 sV[kChunk];               // the round's Nyquist-row values
  __shared__ int8_t sSign[kChunk];
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  const UpsState st = ups[p];
  if (!st.valid) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double2* t1 = T1 + (int64_t)blockIdx.y * n_stored * kUpsNodes;  // [kappa][node]
  const double2* nq = Nq + (int64_t)blockIdx.y * n_stored;
  const int n_kappa = L / 2 + 1;
  // this part's kappa range, of even length (a step takes two kappa); a part past the end sums nothing and writes zeros
  const int per = ((n_kappa + kUpsLongParts - 1) / kUpsLongParts + 1) & ~1;
  const int k_lo = min(n_kappa, (int)blockIdx.x * per), k_hi = min(n_kappa, k_lo + per);
  const int i = lane & 15, k = lane >> 4;
  const int part = k & 1, kk_off = k >> 1;
  const int ct = wave & 1, rt = (wave >> 1) & 1;
  const bool nyq = wave == 4;
  const int ul = nyq ? (lane & 31) : ct * 16 + i;
  const int nyq_half = lane >> 5;
  const double delta_l = (double)st.shift_l + node[ul];
  const double theta = 2.0 * 3.14159265358979323846 * delta_l / (double)L;
  double2 w, step2, jump = mk<double>(1.0, 0.0);
  {
    double sn, cs;
    sincos(theta * (double)(k_lo + (nyq ? nyq_half * (kChunk / 2) : kk_off)), &sn, &cs);
    w = mk<double>(cs, sn);
    sincos((nyq ? 1.0 : 2.0) * theta, &sn, &cs);
    step2 = mk<double>(cs, sn);
    if (nyq) {  // from the end of this half of a round to the same half of the next
      sincos(theta * (double)(kChunk / 2), &sn, &cs);
      jump = mk<double>(cs, sn);
    }
  }
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
  double2 g1 = mk<double>(0.0, 0.0), g2 = mk<double>(0.0, 0.0);
  // staging: element e of a round is (kappa j = e / kUpsNodes, node r = e % kUpsNodes), contiguous in T1
  constexpr int kPer = kChunk * kUpsNodes / kThreads;  // 4 elements per lane and round
  static_assert(kChunk * kUpsNodes % kThreads == 0 && kChunk % 2 == 0 && kChunk <= 64, "staging pattern");
  if (tid < 2 * kUpsNodes) sSelf[tid / kUpsNodes][tid % kUpsNodes] = mk<double>(0.0, 0.0);
  // The loads of the next round are issued (into registers) before the products of this round and completed
  // after them; conjugation and scaling wait until the values are written to LDS.
  double2 pre[kPer];
  KappaEntry pre_k[kPer];
  double2 pre_v = mk<double>(0.0, 0.0);
  KappaEntry pre_vk{0, 1, 1};
  auto fetch = [&](int k0) {
    if (!nyq) {
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int e = tid + q * kThreads, j = e / kUpsNodes, r = e - j * kUpsNodes;
        const int kap = min(k0 + j, n_kappa - 1);
        pre[q] = t1[(int64_t)kap * kUpsNodes + r];
        pre_k[q] = kappa[kap];
      }
    } else if (lane < kChunk) {
      pre_vk = kappa[min(k0 + lane, n_kappa - 1)];
      pre_v = nq[pre_vk.stored];
    }
  };
  auto stage = [&](int k0) {
    if (!nyq) {
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int e = tid + q * kThreads, j = e / kUpsNodes, r = e - j * kUpsNodes;
        const int kap = k0 + j;
        double2 x = mk<double>(0.0, 0.0);
        if (kap < k_hi) {
          const KappaEntry ke = pre_k[q];
          x = pre[q];
          if (ke.sign < 0) x.y = -x.y;
          if (ke.self) {
            const double sgn = ke.sign < 0 ? -1.0 : 1.0;  // Im(Y w) carries the sign of the stored frequency
            sSelf[kap == 0 ? 0 : 1][r] = mk<double>(sgn * x.x, sgn * x.y);
          } else {
            x.x *= 2.0;
            x.y *= 2.0;
          }
        }
        sA[j][r][0] = x.x;
        sA[j][r][1] = -x.y;
      }
    } else if (lane < kChunk) {
      double2 v = pre_v;
      if (pre_vk.sign < 0) v.y = -v.y;
      if (pre_vk.self || k0 + lane >= k_hi) v = mk<double>(0.0, 0.0);  // self-conjugate columns have no partner to correct
      sV[lane] = v;
      sSign[lane] = pre_vk.sign;
    }
  };
  if (k_lo < k_hi) fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += kChunk) {
    __syncthreads();  // the previous round has been consumed
    stage(k0);
    __syncthreads();
    if (k0 + kChunk < k_hi) fetch(k0 + kChunk);  // in flight during the products
    if (!nyq) {
      const int nsteps = (min(kChunk, k_hi - k0) + 1) / 2;
      double nxt = sA[kk_off][16 * rt + i][part];  // the next step's operand is read from LDS while this step's product runs
      for (int sidx = 0; sidx < nsteps; ++sidx) {
        const double av = nxt;
        nxt = sA[min(2 * (sidx + 1), kChunk - 2) + kk_off][16 * rt + i][part];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, part ? w.y : w.x, acc, 0, 0, 0);
        w = rotate(w, step2);
      }
      // (a round cut short by k_hi is the last one: the rotor is not used again)
    } else {
#pragma unroll 4
      for (int j = 0; j < kChunk / 2; ++j) {  // values past k_hi are zero
        const int e = nyq_half * (kChunk / 2) + j;
        const double2 vw = cmul(sV[e], w);
        if (sSign[e] < 0) {
          g1.x += vw.x;
          g1.y += vw.y;
        } else {
          g2.x += vw.x;
          g2.y += vw.y;
        }
        w = cmul(w, step2);
      }
      w = cmul(w, jump);
    }
  }
  __syncthreads();  // sSelf is complete
  if (nyq) {
    double2 g = mk<double>(g1.x + g2.x, g1.y - g2.y);
    g.x += __shfl_xor(g.x, 32);
    g.y += __shfl_xor(g.y, 32);
    if (lane < kUpsNodes) g_part[((int64_t)blockIdx.y * kUpsLongParts + blockIdx.x) * kUpsNodes + ul] = g;
    return;
  }
  // the wave's result tile: column ct * 16 + (lane & 15), rows rt * 16 + (lane >> 4) + 4 q
  double sn, cs;
  sincospi(delta_l, &sn, &cs);  // w at kappa = L / 2 (kappa = 0: w = 1)
  double2* out = ccn_part + ((int64_t)blockIdx.y * kUpsLongParts + blockIdx.x) * kUpsNodes * kUpsNodes;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int us = 16 * rt + k + 4 * q;
    const double2 y0 = sSelf[0][us], yh = sSelf[1][us];
    out[us * kUpsNodes + ul] = mk<double>(acc[q], y0.y + (yh.x * sn + yh.y * cs));
  }
}

// The node values carried to the upsampled grid, CC[us][ul] = sum_js sum_jl P[us][js] C[js][jl] P[ul][jl], and
// the largest |CC|^2 of the block, first in the row-major order of the IMAGE orientation on ties.  C = the sum of the
// parts k_ups_long wrote plus the Nyquist-row term (2 sin(pi ds) g.y, -2 sin(pi ds) g.x), g the summed g_part.
// P[kUpsMax][kUpsNodes] row-major, Pt its transpose.  grid (ceil(n_up / 16), pairs_in_pass): a workgroup owns 16 rows
// of the grid and all its columns.  Both products run on the matrix cores (lane l holds A[l & 15][l >> 4] and
// B[l >> 4][l & 15]; result register g of lane l is C[(l >> 4) + 4 g][l & 15]): first G = P_rows C (wave w: the real
// or imaginary parts, w >> 1, of the node columns 16 (w & 1) ...), then CC = G Pt (wave w: column tiles w, w + 4, w + 8).
__global__ __launch_bounds__(kThreads) void k_ups_interp(const PairDesc* desc, const int8_t* status, int pair0, int S, int L,
                                                         int want_transposed, const double2* __restrict__ ccn_part,
                                                         const double2* __restrict__ g_part,
                                                         const double* __restrict__ node, const double* __restrict__ P,
                                                         const double* __restrict__ Pt,
                                                         const UpsState* __restrict__ ups, int n_up, Peak* partial_ups,
                                                         int n_row_blocks) {
  __shared__ double sC[2][kUpsNodes][kUpsNodes];  // node values [re | im][js][jl]
  __shared__ double sG[2][16][kUpsNodes + 1];     // the block's rows against node columns [re | im][row][jl]
  __shared__ Peak red[kThreads / 64];
  static_assert(kUpsNodes == 32 && kThreads == 256 && kUpsMax <= 12 * 16, "tile assignment below");
  const int p = pair0 + blockIdx.y;
  if (status[p] != MISTITCH_PAIR_OK) return;
  const PairDesc d = desc[p];
  if (d.S != S || d.L != L || (want_transposed >= 0 && d.transposed != want_transposed)) return;
  const UpsState st = ups[p];
  if (!st.valid) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int us0 = blockIdx.x * 16;
  {
    const int jl = tid % kUpsNodes;
    double2 g = mk<double>(0.0, 0.0);
#pragma unroll
    for (int q = 0; q < kUpsLongParts; ++q) {
      const double2 v = g_part[((int64_t)blockIdx.y * kUpsLongParts + q) * kUpsNodes + jl];
      g.x += v.x;
      g.y += v.y;
    }
#pragma unroll
    for (int r = 0; r < kUpsNodes * kUpsNodes / kThreads; ++r) {
      const int js = tid / kUpsNodes + r * (kThreads / kUpsNodes);
      double2 c = mk<double>(0.0, 0.0);
#pragma unroll
      for (int q = 0; q < kUpsLongParts; ++q) {
        const double2 v = ccn_part[((int64_t)blockIdx.y * kUpsLongParts + q) * kUpsNodes * kUpsNodes + js * kUpsNodes + jl];
        c.x += v.x;
        c.y += v.y;
      }
      const double s2 = 2.0 * sinpi((double)st.shift_s + node[js]);
      sC[0][js][jl] = c.x + s2 * g.y;
      sC[1][js][jl] = c.y - s2 * g.x;
    }
  }
  __syncthreads();
  const int i = lane & 15, k = lane >> 4;
  {
    const int comp = wave >> 1, c0 = 16 * (wave & 1);
    const double* prow = P + (int64_t)min(us0 + i, n_up - 1) * kUpsNodes;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < kUpsNodes / 4; ++s4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(prow[4 * s4 + k], sC[comp][4 * s4 + k][c0 + i], acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) sG[comp][k + 4 * g][c0 + i] = acc[g];
  }
  __syncthreads();
  double a_re[kUpsNodes / 4], a_im[kUpsNodes / 4];
#pragma unroll
  for (int s4 = 0; s4 < kUpsNodes / 4; ++s4) {
    a_re[s4] = sG[0][i][4 * s4 + k];
    a_im[s4] = sG[1][i][4 * s4 + k];
  }
  Peak mine{-INFINITY, -1, -1};
  const int n_ct = (n_up + 15) / 16;
  for (int ct = wave; ct < n_ct; ct += kThreads / 64) {
    const int ul = ct * 16 + i;
    const double* pcol = Pt + min(ul, n_up - 1);
    double b[kUpsNodes / 4];
#pragma unroll
    for (int s4 = 0; s4 < kUpsNodes / 4; ++s4) b[s4] = pcol[(4 * s4 + k) * kUpsMax];
    d4 cr = d4{0.0, 0.0, 0.0, 0.0}, ci = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < kUpsNodes / 4; ++s4) {
      cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a_re[s4], b[s4], cr, 0, 0, 0);
      ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im[s4], b[s4], ci, 0, 0, 0);
    }
    if (ul < n_up) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int us = us0 + k + 4 * g;
        if (us < n_up) {
          const double mag2 = cr[g] * cr[g] + ci[g] * ci[g];
          const int key = d.transposed ? ul * n_up + us : us * n_up + ul;
          if (first_better(mag2, key, mine.val, mine.key)) mine = Peak{mag2, key, 0};
        }
      }
    }
  }
  const Peak best = block_argmax_first(mine, red);
  if (tid == 0) partial_ups[(int64_t)blockIdx.y * n_row_blocks + blockIdx.x] = best;
}

// One thread per pair: the refined shift and its rounded position as the last candidate slot.
__global__ void k_ups_finish(const PairDesc* desc, const int8_t* status, int pair0, int n, const Peak* partial_ups,
                             int n_row_blocks, int n_up, double uf, UpsState* ups, Cand* cand, int n_cand) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = pair0 + i;
  if (status[p] != MISTITCH_PAIR_OK) return;
  UpsState st = ups[p];
  if (!st.valid) return;
  const PairDesc d = desc[p];
  Peak best{-INFINITY, -1, -1};
  for (int b = 0; b < n_row_blocks; ++b) {
    const Peak c = partial_ups[(int64_t)i * n_row_blocks + b];
    if (first_better(c.val, c.key, best.val, best.key)) best = c;
  }
  if (best.key < 0) {
    st.valid = 0;
    ups[p] = st;
    return;
  }
  const int uy = best.key / n_up, ux = best.key - uy * n_up;
  const int shift_y = d.transposed ? st.shift_l : st.shift_s, shift_x = d.transposed ? st.shift_s : st.shift_l;
  const double sy = (double)shift_y + ((double)uy - (double)(n_up / 2)) / uf;
  const double sx = (double)shift_x + ((double)ux - (double)(n_up / 2)) / uf;
  st.sub_s = d.transposed ? sx : sy;
  st.sub_l = d.transposed ? sy : sx;
  ups[p] = st;
  Cand c{(int)rint(sy), (int)rint(sx), 1, -1};
  const int ay = c.y < 0 ? -c.y : c.y, ax = c.x < 0 ? -c.x : c.x;
  if (ay >= d.img_h || ax >= d.img_w)
    c.flag = 1;
  else if ((long long)(d.img_h - ay) * (d.img_w - ax) < 25)
    c.flag = 2;
  else
    c.flag = 0;
  cand[(int64_t)p * (n_cand + 1) + n_cand] = c;
}

// Exact integer moments of two uint16 rows, 8 pixels per lane and step.  Pixels
// are split into byte planes so that every product is a v_dot4_u32_u8.  The busiest
// plane (ab_x: two dot products of at most 4 * 255 * 255 per add4) wraps 32 bits after
// 8256 calls of add4 by one lane: k_ncc_accumulate is bounded by the strip limits
// (at most 512 rows x 2 groups x 2 calls per lane), the callers without such a bound use
// moments_of_rows<true>, which empties the planes into 64-bit sums long before that.
struct __attribute__((packed, aligned(2))) U16x8 {
  unsigned d[4];
};

struct MomentAcc {
  // byte-plane partial sums: l = low bytes, h = high bytes
  unsigned a_l = 0, a_h = 0, b_l = 0, b_h = 0;
  unsigned aa_ll = 0, aa_lh = 0, aa_hh = 0, bb_ll = 0, bb_lh = 0, bb_hh = 0;
  unsigned ab_ll = 0, ab_x = 0, ab_hh = 0;
  __device__ __forceinline__ void add4(unsigned a0, unsigned a1, unsigned b0, unsigned b1) {
    const unsigned al = __builtin_amdgcn_perm(a1, a0, 0x06040200u), ah = __builtin_amdgcn_perm(a1, a0, 0x07050301u);
    const unsigned bl = __builtin_amdgcn_perm(b1, b0, 0x06040200u), bh = __builtin_amdgcn_perm(b1, b0, 0x07050301u);
    a_l = __builtin_amdgcn_udot4(al, 0x01010101u, a_l, false);
    a_h = __builtin_amdgcn_udot4(ah, 0x01010101u, a_h, false);
    b_l = __builtin_amdgcn_udot4(bl, 0x01010101u, b_l, false);
    b_h = __builtin_amdgcn_udot4(bh, 0x01010101u, b_h, false);
    aa_ll = __builtin_amdgcn_udot4(al, al, aa_ll, false);
    aa_lh = __builtin_amdgcn_udot4(al, ah, aa_lh, false);
    aa_hh = __builtin_amdgcn_udot4(ah, ah, aa_hh, false);
    bb_ll = __builtin_amdgcn_udot4(bl, bl, bb_ll, false);
    bb_lh = __builtin_amdgcn_udot4(bl, bh, bb_lh, false);
    bb_hh = __builtin_amdgcn_udot4(bh, bh, bb_hh, false);
    ab_ll = __builtin_amdgcn_udot4(al, bl, ab_ll, false);
    ab_x = __builtin_amdgcn_udot4(al, bh, ab_x, false);
    ab_x = __builtin_amdgcn_udot4(ah, bl, ab_x, false);
    ab_hh = __builtin_amdgcn_udot4(ah, bh, ab_hh, false);
  }
  __device__ __forceinline__ void add8(const U16x8& a, const U16x8& b) {
    add4(a.d[0], a.d[1], b.d[0], b.d[1]);
    add4(a.d[2], a.d[3], b.d[2], b.d[3]);
  }
  // one pixel pair
  __device__ __forceinline__ void add1(unsigned a, unsigned b) { add4(a, 0u, b, 0u); }
  // A call of add4 adds at most 2 * 4 * 255 * 255 = 520,200 to any plane, so after n calls per
  // lane the sum of a plane over the 64 lanes of a wave fits 32 bits while n <= kWaveSteps32.
  static constexpr int kWaveSteps32 = 128;
  // Moments summed over the wave, the same in every lane; n = calls of add4 by the busiest lane.
  __device__ __forceinline__ void wave_totals(int n, unsigned long long (&v)[5]) const {
    if (n <= kWaveSteps32) {
      v[0] = wave_sum_u32(a_l + (a_h << 8));
      v[1] = wave_sum_u32(b_l + (b_h << 8));
      v[2] = (unsigned long long)wave_sum_u32(aa_ll) + ((unsigned long long)wave_sum_u32(aa_lh) << 9) +
             ((unsigned long long)wave_sum_u32(aa_hh) << 16);
      v[3] = (unsigned long long)wave_sum_u32(bb_ll) + ((unsigned long long)wave_sum_u32(bb_lh) << 9) +
             ((unsigned long long)wave_sum_u32(bb_hh) << 16);
      v[4] = (unsigned long long)wave_sum_u32(ab_ll) + ((unsigned long long)wave_sum_u32(ab_x) << 8) +
             ((unsigned long long)wave_sum_u32(ab_hh) << 16);
    } else {
      totals(v);
#pragma unroll
      for (int i = 0; i < 5; ++i) v[i] = wave_sum_u64(v[i]);
    }
  }
  // v[0..5) = sum a, sum b, sum a^2, sum b^2, sum ab of this lane
  __device__ __forceinline__ void totals(unsigned long long (&v)[5]) const {
    v[0] = (unsigned long long)a_l + ((unsigned long long)a_h << 8);
    v[1] = (unsigned long long)b_l + ((unsigned long long)b_h << 8);
    v[2] = (unsigned long long)aa_ll + ((unsigned long long)aa_lh << 9) + ((unsigned long long)aa_hh << 16);
    v[3] = (unsigned long long)bb_ll + ((unsigned long long)bb_lh << 9) + ((unsigned long long)bb_hh << 16);
    v[4] = (unsigned long long)ab_ll + ((unsigned long long)ab_x << 8) + ((unsigned long long)ab_hh << 16);
  }
};

// Rows [r_begin, r_end) step r_step of two windows of ncols pixels; all 256 threads.  Moments
// summed over each wave come back in v (the same in every lane of a wave).
// FLUSH: callers whose shapes are not bounded (any patch, any edge width) empty a lane's 32-bit
// byte-plane sums into 64-bit ones every kFlushEvery groups, long before the busiest plane
// (two dot products of at most 4 * 255 * 255 per call of add4) can wrap at 8256 calls.
constexpr int kFlushEvery = 2048;  // calls of add8 = 4096 of add4
template <bool FLUSH = false>
__device__ __forceinline__ void moments_of_rows(const uint16_t* a, int64_t pa, const uint16_t* b, int64_t pb,
                                                int r_begin, int r_end, int r_step, int ncols,
                                                unsigned long long (&v)[5]) {
  MomentAcc acc;
  unsigned long long wide[5] = {0, 0, 0, 0, 0};
  int since = 0;
  auto flush = [&]() {
    if constexpr (FLUSH) {
      if (++since >= kFlushEvery) {
        unsigned long long t[5];
        acc.totals(t);
#pragma unroll
        for (int i = 0; i < 5; ++i) wide[i] += t[i];
        acc = MomentAcc();
        since = 0;
      }
    }
  };
  const int ngroups = ncols >> 3;
  // The columns beyond the last whole group are taken one pixel pair per thread and step, eight
  // threads to a row, apart from the rows' trips; the first step's pixels (all of them, when a
  // workgroup has no more than 32 rows) are requested here, ahead of the groups, and used last.
  const int ntail = ncols & 7;
  const bool tail_lane = ((int)threadIdx.x & 7) < ntail;
  const int64_t tail_col = 8 * ngroups + ((int)threadIdx.x & 7);
  const int tail_row = r_begin + ((int)threadIdx.x >> 3) * r_step;
  const bool tail_first = tail_lane && tail_row < r_end;
  unsigned tail_a = 0, tail_b = 0;
  if (tail_first) {
    tail_a = a[(int64_t)tail_row * pa + tail_col];
    tail_b = b[(int64_t)tail_row * pb + tail_col];
  }
  int r = r_begin;
  {
    // two rows per trip: four 16-byte loads in flight per lane (three or four rows gain nothing more)
    constexpr int kRows = 2;
    for (; r + (kRows - 1) * r_step < r_end; r += kRows * r_step) {
      const uint16_t* ra = a + (int64_t)r * pa;
      const uint16_t* rb = b + (int64_t)r * pb;
      for (int g = threadIdx.x; g < ngroups; g += kThreads) {
        U16x8 x[kRows], y[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
          x[k] = *reinterpret_cast<const U16x8*>(ra + (int64_t)k * r_step * pa + 8 * g);
          y[k] = *reinterpret_cast<const U16x8*>(rb + (int64_t)k * r_step * pb + 8 * g);
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
          acc.add8(x[k], y[k]);
          flush();
        }
      }
    }
  }
  for (; r < r_end; r += r_step) {
    const uint16_t* ra = a + (int64_t)r * pa;
    const uint16_t* rb = b + (int64_t)r * pb;
    for (int g = threadIdx.x; g < ngroups; g += kThreads) {
      acc.add8(*reinterpret_cast<const U16x8*>(ra + 8 * g), *reinterpret_cast<const U16x8*>(rb + 8 * g));
      flush();
    }
  }
  if (tail_first) {
    acc.add1(tail_a, tail_b);
    for (int rr = tail_row + (kThreads >> 3) * r_step; rr < r_end; rr += (kThreads >> 3) * r_step) {
      acc.add1(a[(int64_t)rr * pa + tail_col], b[(int64_t)rr * pb + tail_col]);
      flush();
    }
  }
  const int nrows = r_end > r_begin ? (r_end - r_begin + r_step - 1) / r_step : 0;
  const int steps = 2 * nrows * ((ngroups + kThreads - 1) / kThreads) +
                    (ntail ? (nrows + (kThreads >> 3) - 1) / (kThreads >> 3) : 0);
  if constexpr (FLUSH) {
    if (steps > MomentAcc::kWaveSteps32) {  // wave-uniform; below that nothing has been flushed
      unsigned long long t[5];
      acc.totals(t);
#pragma unroll
      for (int i = 0; i < 5; ++i) v[i] = wave_sum_u64(wide[i] + t[i]);
      return;
    }
  }
  acc.wave_totals(steps, v);
}

// Sum the wave totals v[0..5) over the workgroup and add them to acc[0..5) with integer
// atomics (exact and order independent).
__device__ __forceinline__ void reduce_moments(unsigned long long (&v)[5], unsigned long long* red,
                                               unsigned long long* acc) {
  // v holds wave totals.  One set of atomics per workgroup: four times as many (one per wave) contend enough on the
  // candidate's five words to cost more than this barrier
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) red[wave * 5 + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    unsigned long long t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += red[w * 5 + threadIdx.x];
    atomicAdd(&acc[threadIdx.x], t);
  }
}

// Exact moments (sum a, sum b, sum a^2, sum b^2, sum ab) over the overlap of a
// candidate shift.  One-dimensional grid of 8 * ceil(pairs_in_pass / 8) * (n_cand + 1) * kNccChunks
// workgroups.  Workgroups go to the eight XCDs in turn, each with an L2 of its own, so
// workgroup b works for pair (b mod 8) + 8 * (...): every candidate of a pair is then summed
// on one XCD, whose L2 keeps that pair's two strips while its ~30 overlaps are read from them.
constexpr int kXcds = 8;
constexpr int kShortStripRows = 256;  // the usual edge width: sizes the "small overlap" test below
// Candidates [ci0, ci0 + n_ci) of every pair are summed: all n_cand + 1 of them, or the subpixel slot alone when
// k_ncc_sweep has taken the signed interpretations of the peaks.
__global__ __launch_bounds__(kThreads, 8) void k_ncc_accumulate(const PairDesc* desc, const int8_t* status,
                                                             int pair0, int n_in_pass, const Cand* cand,
                                                             unsigned long long* acc, int n_cand, int ci0, int n_ci) {
  __shared__ unsigned long long red[5 * kThreads / 64];
  const int xcd = blockIdx.x % kXcds, seq = blockIdx.x / kXcds;
  const int slots = n_cand + 1;
  const int kPerPair = kNccChunks * n_ci;
  const int pz = xcd + kXcds * (seq / kPerPair);
  if (pz >= n_in_pass) return;
  const int within = seq % kPerPair;
  const int ci = ci0 + within / kNccChunks, chunk = within % kNccChunks;
  const int p = pair0 + pz;
  // three independent lookups, requested together ahead of the first test
  const int8_t st = status[p];
  const Cand c = cand[(int64_t)p * slots + ci];
  const PairDesc d = desc[p];
  if (st != MISTITCH_PAIR_OK || c.flag != 0) return;
  const int ds = d.transposed ? c.x : c.y;
  const int dl = d.transposed ? c.y : c.x;
  const int s_lo = ds > 0 ? ds : 0, s_hi = d.S + (ds < 0 ? ds : 0);
  const int l_lo = dl > 0 ? dl : 0, l_hi = d.L + (dl < 0 ? dl : 0);
  int first = s_lo + chunk, step = kNccChunks;
  // an overlap of no more work than one workgroup's share of a full one is not worth the set-up
  // and the reduction of eight: the first workgroup takes all its rows
  if ((int64_t)(s_hi - s_lo) * ((l_hi - l_lo + 7) >> 3) <= (int64_t)kThreads * (kShortStripRows / kNccChunks)) {
    if (chunk != 0) return;
    first = s_lo;
    step = 1;
  }
  if (first >= s_hi) return;
  unsigned long long v[5];
  moments_of_rows(d.a + l_lo, d.pitch, d.b + (int64_t)(-ds) * d.pitch + (l_lo - dl), d.pitch, first, s_hi, step,
                  l_hi - l_lo, v);
  reduce_moments(v, red, acc + ((int64_t)p * slots + ci) * 5);
}

// ---- candidate moments by circular sweeps ----------------------------------------
// The 16 signed interpretations of a peak (n rows, m columns in canonical orientation) are the shifts
// (+-n | +-(S - n), +-m | +-(L - m)).  Shifts +n and -(S - n) pair row r of the first strip with row (r - n) mod S of
// the second: the rows r >= n belong to the former, the rows r < n to the latter, and together they are every row
// once.  The same holds along a row for +m and -(L - m), and for the opposite lags.  So four circular sweeps of the
// strip pair (row lag +-n, column lag +-m) visit every pixel pair of all 16 overlaps exactly once, where 16 separate
// windows read four strip areas of either strip.  A workgroup takes up to kSweepRows rows that lie on one side of the
// wrap row of EITHER row lag, loads a row of the first strip ONCE (16 bytes per lane, aligned) and its partner rows of
// both row lags at the two column lags (four loads, on 4-byte boundaries; a row lag whose candidates are all ruled out
// is left out), and keeps the byte-plane sums of all four: the moments of the first strip are formed once, not
// once per candidate.  Every lane owns one 8-pixel group of the row, which lies on one side of a lag's wrap column for
// all rows, so its sums go to one candidate per row lag and column lag.  The one group that a column lag's wrap column
// cuts is left out of that lag's sums and taken pixel by pixel afterwards, wave w for (lag w / 2, side w % 2).
// Byte-plane sums are reduced over the 16 lanes of a DPP row in 32 bits and added to the workgroup's 64-bit planes in
// LDS; 40 threads turn the planes into moments and add them to the candidates' accumulators.
// grid: 8 * ceil(pairs_in_pass / 8) * n_peaks * n_row_chunks * n_col_chunks workgroups, a pair's on one XCD as in
// k_ncc_accumulate.  Rows are 16-byte aligned and L is a multiple of 8 (the host checks).
#ifndef MS_SWEEP_ROWS
#define MS_SWEEP_ROWS 64
#endif
#ifndef MS_SWEEP_OCC
#define MS_SWEEP_OCC 5
#endif
constexpr int kSweepRows = MS_SWEEP_ROWS;
// a lane adds four pixels to a stream twice per row; the busiest byte plane gains at most 520,200 each time, and the
// sum of a plane over the 16 lanes of a DPP row has to fit 32 bits (row_sum)
static_assert(2ll * kSweepRows * 520200ll * 16ll < (1ll << 32), "byte-plane sums of 16 lanes overflow 32 bits");
constexpr int kPlanes = 13;  // a_l a_h b_l b_h aa_ll aa_lh aa_hh bb_ll bb_lh bb_hh ab_ll ab_x ab_hh

// Sum of v over each 16 consecutive lanes, valid in every lane of the 16.
__device__ __forceinline__ unsigned row_sum(unsigned v) {
  v += dpp_u32<kDppXor1>(v);
  v += dpp_u32<kDppXor2>(v);
  v += dpp_u32<kDppHalfMirror>(v);
  v += dpp_u32<kDppMirror>(v);
  return v;
}

// NB second-strip streams against one first-strip stream (2: the column lags of one row lag; 4: of both row lags,
// stream 2 * row lag + column lag).
template <int NB>
struct SweepAcc {
  unsigned a_l = 0, a_h = 0, aa_ll = 0, aa_lh = 0, aa_hh = 0;  // first strip: shared by all streams
  unsigned b_l[NB] = {}, b_h[NB] = {}, bb_ll[NB] = {}, bb_lh[NB] = {}, bb_hh[NB] = {};
  unsigned ab_ll[NB] = {}, ab_x[NB] = {}, ab_hh[NB] = {};
  // four pixels of the first strip: their own sums; al, ah = their low and high bytes
  __device__ __forceinline__ void first(unsigned a0, unsigned a1, unsigned& al, unsigned& ah) {
    al = __builtin_amdgcn_perm(a1, a0, 0x06040200u), ah = __builtin_amdgcn_perm(a1, a0, 0x07050301u);
    a_l = __builtin_amdgcn_udot4(al, 0x01010101u, a_l, false);
    a_h = __builtin_amdgcn_udot4(ah, 0x01010101u, a_h, false);
    aa_ll = __builtin_amdgcn_udot4(al, al, aa_ll, false);
    aa_lh = __builtin_amdgcn_udot4(al, ah, aa_lh, false);
    aa_hh = __builtin_amdgcn_udot4(ah, ah, aa_hh, false);
  }
  // the four pixels that stream k pairs with them
  __device__ __forceinline__ void second(int k, unsigned al, unsigned ah, unsigned b0, unsigned b1) {
    const unsigned bl = __builtin_amdgcn_perm(b1, b0, 0x06040200u), bh = __builtin_amdgcn_perm(b1, b0, 0x07050301u);
    b_l[k] = __builtin_amdgcn_udot4(bl, 0x01010101u, b_l[k], false);
    b_h[k] = __builtin_amdgcn_udot4(bh, 0x01010101u, b_h[k], false);
    bb_ll[k] = __builtin_amdgcn_udot4(bl, bl, bb_ll[k], false);
    bb_lh[k] = __builtin_amdgcn_udot4(bl, bh, bb_lh[k], false);
    bb_hh[k] = __builtin_amdgcn_udot4(bh, bh, bb_hh[k], false);
    ab_ll[k] = __builtin_amdgcn_udot4(al, bl, ab_ll[k], false);
    ab_x[k] = __builtin_amdgcn_udot4(al, bh, ab_x[k], false);
    ab_x[k] = __builtin_amdgcn_udot4(ah, bl, ab_x[k], false);
    ab_hh[k] = __builtin_amdgcn_udot4(ah, bh, ab_hh[k], false);
  }
  // streams K0 (if LA) and K0 + 1 (if LB)
  template <bool LA, bool LB, int K0 = 0>
  __device__ __forceinline__ void add4(unsigned a0, unsigned a1, unsigned x0, unsigned x1, unsigned y0, unsigned y1) {
    unsigned al, ah;
    first(a0, a1, al, ah);
    if constexpr (LA) second(K0, al, ah, x0, x1);
    if constexpr (LB) second(K0 + 1, al, ah, y0, y1);
  }
  __device__ __forceinline__ void planes(int k, unsigned (&v)[kPlanes]) const {
    v[0] = a_l, v[1] = a_h, v[2] = b_l[k], v[3] = b_h[k], v[4] = aa_ll, v[5] = aa_lh, v[6] = aa_hh;
    v[7] = bb_ll[k], v[8] = bb_lh[k], v[9] = bb_hh[k], v[10] = ab_ll[k], v[11] = ab_x[k], v[12] = ab_hh[k];
  }
};

// The planes of the lanes with `take` set, summed over each DPP row and added to dst[0..13) (LDS).
__device__ __forceinline__ void add_planes(const unsigned (&v)[kPlanes], bool take, unsigned long long* dst) {
  const bool writer = (threadIdx.x & 15) == 0;
#pragma unroll
  for (int i = 0; i < kPlanes; ++i) {
    const unsigned t = row_sum(take ? v[i] : 0u);
    if (writer && t != 0u) atomicAdd(&dst[i], (unsigned long long)t);
  }
}

// Lane i takes the value of lane i + 1 of the wave (the last lane keeps its own).
constexpr int kDppWaveShl1 = 0x130;
__device__ __forceinline__ unsigned from_next_lane(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(v, v, kDppWaveShl1, 0xf, 0xf, false);
}

// The 8 pixels from an ODD pixel offset on: x holds the four dwords from one pixel before (a 4-byte aligned load: a
// 16-byte load from an odd 2-byte boundary costs about twice one from a 4-byte boundary), the eighth pixel is the low half of the
// dword behind them, which is the first dword of the next lane's load, or `own_tail` for a lane with `lonely` set.
__device__ __forceinline__ void shift_in_odd(U16x8& x, unsigned own_tail, bool lonely) {
  const unsigned next = from_next_lane(x.d[0]);
  const unsigned d4 = lonely ? own_tail : next;
  x.d[0] = __builtin_amdgcn_alignbit(x.d[1], x.d[0], 16);
  x.d[1] = __builtin_amdgcn_alignbit(x.d[2], x.d[1], 16);
  x.d[2] = __builtin_amdgcn_alignbit(x.d[3], x.d[2], 16);
  x.d[3] = __builtin_amdgcn_alignbit(d4, x.d[3], 16);
}

// ODD: both column lags are odd (off_a and off_b of every lane whose sums count are odd; lanes that do not count
// have offset 0).  lonely_a / lonely_b: the lane after this one does not load the pixels that follow this lane's.
template <bool LA, bool LB, bool ODD, int K0>
__device__ __forceinline__ void sweep_rows(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, int64_t pitch,
                                           int r0, int r1, int off_a, int off_b, bool lonely_a, bool lonely_b,
                                           SweepAcc<4>& acc) {
  // a, b: the lane's group in row 0 of the first strip / the start of the paired row of the second
  const int ld_a = ODD ? max(off_a - 1, 0) : off_a, ld_b = ODD ? max(off_b - 1, 0) : off_b;
  const bool lonely = ODD && ((LA && lonely_a) || (LB && lonely_b));
  int r = r0;
#pragma unroll 1
  for (; r + 1 < r1; r += 2) {  // two rows per trip: six 16-byte loads in flight per lane
    const uint16_t* ar = a + (int64_t)r * pitch;
    const uint16_t* br = b + (int64_t)r * pitch;
    const U16x8 va0 = *reinterpret_cast<const U16x8*>(ar), va1 = *reinterpret_cast<const U16x8*>(ar + pitch);
    U16x8 xa0{}, xa1{}, xb0{}, xb1{};
#if defined(MS_SWEEP_ABLATE) && MS_SWEEP_ABLATE == 1  /* measurement build: the second strip is not loaded */
    xa0 = xb0 = va1;
    xa1 = xb1 = va0;
    (void)br;
#else
    if constexpr (LA) {
      xa0 = *reinterpret_cast<const U16x8*>(br + ld_a);
      xa1 = *reinterpret_cast<const U16x8*>(br + pitch + ld_a);
    }
    if constexpr (LB) {
      xb0 = *reinterpret_cast<const U16x8*>(br + ld_b);
      xb1 = *reinterpret_cast<const U16x8*>(br + pitch + ld_b);
    }
    if constexpr (ODD) {
      unsigned ta0 = 0, ta1 = 0, tb0 = 0, tb1 = 0;
      if (lonely) {  // few lanes: the last of a wave, the one before the group that the wrap column cuts
        if constexpr (LA) ta0 = br[off_a + 7], ta1 = br[pitch + off_a + 7];
        if constexpr (LB) tb0 = br[off_b + 7], tb1 = br[pitch + off_b + 7];
      }
      if constexpr (LA) shift_in_odd(xa0, ta0, lonely_a), shift_in_odd(xa1, ta1, lonely_a);
      if constexpr (LB) shift_in_odd(xb0, tb0, lonely_b), shift_in_odd(xb1, tb1, lonely_b);
    }
#endif
#if defined(MS_SWEEP_ABLATE) && MS_SWEEP_ABLATE == 2  /* measurement build: loads only */
    acc.a_l ^= va0.d[0] ^ va0.d[1] ^ va0.d[2] ^ va0.d[3] ^ va1.d[0] ^ va1.d[1] ^ va1.d[2] ^ va1.d[3] ^ xa0.d[0] ^ xa0.d[1] ^
               xa0.d[2] ^ xa0.d[3] ^ xb0.d[0] ^ xb0.d[1] ^ xb0.d[2] ^ xb0.d[3] ^ xa1.d[0] ^ xa1.d[1] ^ xa1.d[2] ^ xa1.d[3] ^
               xb1.d[0] ^ xb1.d[1] ^ xb1.d[2] ^ xb1.d[3];
#else
    acc.add4<LA, LB, K0>(va0.d[0], va0.d[1], xa0.d[0], xa0.d[1], xb0.d[0], xb0.d[1]);
    acc.add4<LA, LB, K0>(va0.d[2], va0.d[3], xa0.d[2], xa0.d[3], xb0.d[2], xb0.d[3]);
    acc.add4<LA, LB, K0>(va1.d[0], va1.d[1], xa1.d[0], xa1.d[1], xb1.d[0], xb1.d[1]);
    acc.add4<LA, LB, K0>(va1.d[2], va1.d[3], xa1.d[2], xa1.d[3], xb1.d[2], xb1.d[3]);
#endif
  }
  if (r < r1) {
    const uint16_t* ar = a + (int64_t)r * pitch;
    const uint16_t* br = b + (int64_t)r * pitch;
    const U16x8 va0 = *reinterpret_cast<const U16x8*>(ar);
    U16x8 xa0{}, xb0{};
    if constexpr (LA) xa0 = *reinterpret_cast<const U16x8*>(br + ld_a);
    if constexpr (LB) xb0 = *reinterpret_cast<const U16x8*>(br + ld_b);
    if constexpr (ODD) {
      unsigned ta0 = 0, tb0 = 0;
      if (lonely) {
        if constexpr (LA) ta0 = br[off_a + 7];
        if constexpr (LB) tb0 = br[off_b + 7];
      }
      if constexpr (LA) shift_in_odd(xa0, ta0, lonely_a);
      if constexpr (LB) shift_in_odd(xb0, tb0, lonely_b);
    }
    acc.add4<LA, LB, K0>(va0.d[0], va0.d[1], xa0.d[0], xa0.d[1], xb0.d[0], xb0.d[1]);
    acc.add4<LA, LB, K0>(va0.d[2], va0.d[3], xa0.d[2], xa0.d[3], xb0.d[2], xb0.d[3]);
  }
}

// Both row lags at once: row r of the first strip is loaded once and meets its partner rows of both lags (b0: lag +n,
// b1: lag -n), each at the two column lags.  One row per trip: five 16-byte loads in flight per lane.
template <bool LA, bool LB, bool ODD>
__device__ __forceinline__ void sweep_rows_both(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b0,
                                                const uint16_t* __restrict__ b1, int64_t pitch, int r0, int r1, int off_a,
                                                int off_b, bool lonely_a, bool lonely_b, SweepAcc<4>& acc) {
  const int ld_a = ODD ? max(off_a - 1, 0) : off_a, ld_b = ODD ? max(off_b - 1, 0) : off_b;
  const bool lonely = ODD && ((LA && lonely_a) || (LB && lonely_b));
#pragma unroll 1
  for (int r = r0; r < r1; ++r) {
    const uint16_t* ar = a + (int64_t)r * pitch;
    const uint16_t* br0 = b0 + (int64_t)r * pitch;
    const uint16_t* br1 = b1 + (int64_t)r * pitch;
    const U16x8 va = *reinterpret_cast<const U16x8*>(ar);
    U16x8 xa0{}, xb0{}, xa1{}, xb1{};
    if constexpr (LA) {
      xa0 = *reinterpret_cast<const U16x8*>(br0 + ld_a);
      xa1 = *reinterpret_cast<const U16x8*>(br1 + ld_a);
    }
    if constexpr (LB) {
      xb0 = *reinterpret_cast<const U16x8*>(br0 + ld_b);
      xb1 = *reinterpret_cast<const U16x8*>(br1 + ld_b);
    }
    if constexpr (ODD) {
      unsigned ta0 = 0, ta1 = 0, tb0 = 0, tb1 = 0;
      if (lonely) {
        if constexpr (LA) ta0 = br0[off_a + 7], ta1 = br1[off_a + 7];
        if constexpr (LB) tb0 = br0[off_b + 7], tb1 = br1[off_b + 7];
      }
      if constexpr (LA) shift_in_odd(xa0, ta0, lonely_a), shift_in_odd(xa1, ta1, lonely_a);
      if constexpr (LB) shift_in_odd(xb0, tb0, lonely_b), shift_in_odd(xb1, tb1, lonely_b);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned al, ah;
      acc.first(va.d[2 * h], va.d[2 * h + 1], al, ah);
      if constexpr (LA) {
        acc.second(0, al, ah, xa0.d[2 * h], xa0.d[2 * h + 1]);
        acc.second(2, al, ah, xa1.d[2 * h], xa1.d[2 * h + 1]);
      }
      if constexpr (LB) {
        acc.second(1, al, ah, xb0.d[2 * h], xb0.d[2 * h + 1]);
        acc.second(3, al, ah, xb1.d[2 * h], xb1.d[2 * h + 1]);
      }
    }
  }
}

__global__ __launch_bounds__(kThreads, MS_SWEEP_OCC) void k_ncc_sweep(const PairDesc* desc, const int8_t* status, int pair0,
                                                                      int n_in_pass, const Peak* peaks, const Cand* cand,
                                                                      unsigned long long* acc, int n_cand, int n_peaks,
                                                                      int n_row_chunks, int n_col_chunks, int block_rows) {
  __shared__ unsigned long long red[8][kPlanes + 1];  // [4 * row lag + 2 * column lag + side of the wrap column][byte plane]
  __shared__ unsigned long long red2[5 * kThreads / 64];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x % kXcds, seq = blockIdx.x / kXcds;
  const int per_pair = n_peaks * n_row_chunks * n_col_chunks;
  const int pz = xcd + kXcds * (seq / per_pair);
  if (pz >= n_in_pass) return;
  // a pair's workgroups in row order: the sweeps of all peaks over the same rows of the first strip are neighbours in
  // the grid and read those rows from L2 at about the same time
  int within = seq % per_pair;
  const int cc = within % n_col_chunks;
  within /= n_col_chunks;
  const int pk = within % n_peaks, ch = within / n_peaks;
  const int p = pair0 + pz;
  const int8_t st = status[p];
  const Peak peak = peaks[(int64_t)p * kMaxK + pk];
  const PairDesc d = desc[p];
  if (st != MISTITCH_PAIR_OK || peak.key < 0) return;
  const int py = peak.key / d.img_w, px = peak.key - py * d.img_w;
  const int n = d.transposed ? px : py, m = d.transposed ? py : px;  // canonical rows, columns: 0 <= n < S, 0 <= m < L
  const int S = d.S, L = d.L;
  // rows: lag +n (sig 0) pairs row r with row r - n (r >= n: shift +n) or r - n + S (r < n: shift -(S - n)); lag -n
  // (sig 1) pairs it with r + n (r < S - n: shift -n) or r + n - S (shift +(S - n)).  Their wrap rows are n and S - n.
  // Row chunks: the fixed blocks [j R, (j + 1) R) of the strip (R = block_rows <= kSweepRows), cut at both wrap rows,
  // in row order: every chunk lies on one side of either wrap row.  boundary(i): the i-th smallest of the block starts and the wrap rows.
  const int w1 = min(n, S - n), w2 = max(n, S - n);
  const bool e1 = w1 % block_rows != 0, e2 = w2 % block_rows != 0 && w2 != w1;
  auto boundary = [&](int i) {
    if (e2) {
      const int i2 = w2 / block_rows + 1 + (e1 ? 1 : 0);
      if (i == i2) return w2;
      if (i > i2) --i;
    }
    if (e1) {
      const int i1 = w1 / block_rows + 1;
      if (i == i1) return w1;
      if (i > i1) --i;
    }
    return min(i * block_rows, S);
  };
  const int r0 = boundary(ch), r1 = boundary(ch + 1);
  if (r0 >= r1) return;
  // the eight candidates of this workgroup, 4 * sig + q with q = 2 * lag + side: lag 0 reads the second strip at column
  // (c - m) mod L (side 1, c >= m: shift +m; side 0: shift -(L - m)), lag 1 at (c + m) mod L (side 0, c < L - m:
  // shift -m; side 1: shift +(L - m))
  const int slots = n_cand + 1;
  int ci[8], dl[4], ds[2];
  bool live[8];
#pragma unroll
  for (int sig = 0; sig < 2; ++sig) {
    const bool below = r0 < (sig ? S - n : n);
    const int row_mag = (sig != 0) != below ? 1 : 0;  // 0: magnitude n, 1: S - n
    const int row_pos = below ? 0 : 1;                // sign of the row shift
    ds[sig] = (row_pos ? 1 : -1) * (row_mag ? S - n : n);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lag = q >> 1, side = q & 1;
      const int col_mag = lag ? side : 1 - side;
      const int col_pos = side;
      dl[q] = (col_pos ? 1 : -1) * (col_mag ? L - m : m);
      const int t = d.transposed ? (col_mag << 3) | (row_mag << 2) | (col_pos << 1) | row_pos
                                 : (row_mag << 3) | (col_mag << 2) | (row_pos << 1) | col_pos;
      ci[4 * sig + q] = 16 * pk + t;
      live[4 * sig + q] = cand[(int64_t)p * slots + ci[4 * sig + q]].flag == 0;
    }
  }
  const bool sig_live[2] = {live[0] || live[1] || live[2] || live[3], live[4] || live[5] || live[6] || live[7]};
  const bool lag_a = live[0] | live[1] | live[4] | live[5], lag_b = live[2] | live[3] | live[6] | live[7];
  if (!(lag_a | lag_b)) return;
  const int ngroups = L >> 3;
  const int g0 = cc * kThreads;  // first group of this column chunk
  // the groups the wrap columns cut (-1: none, or not in this chunk, or a lag without a live candidate)
  int cut_a = (lag_a && (m & 7)) ? (m >> 3) : -1;
  int cut_b = (lag_b && ((L - m) & 7)) ? ((L - m) >> 3) : -1;
  if (cut_a >= 0 && cut_a == cut_b) {
    // both wrap columns cut the same group (m within 8 of L / 2): the overlaps one by one, windowed as
    // k_ncc_accumulate does, by the workgroups of the first column chunk
    if (cc != 0) return;
#pragma unroll 1
    for (int e = 0; e < 8; ++e) {
      if (!live[e]) continue;  // uniform
      const int q = e & 3, row_shift = ds[e >> 2];
      const int l_lo = dl[q] > 0 ? dl[q] : 0, l_hi = L + (dl[q] < 0 ? dl[q] : 0);
      unsigned long long v[5];
      moments_of_rows(d.a + l_lo, d.pitch, d.b + (int64_t)(-row_shift) * d.pitch + (l_lo - dl[q]), d.pitch, r0, r1, 1,
                      l_hi - l_lo, v);
      __syncthreads();
      reduce_moments(v, red2, acc + ((int64_t)p * slots + ci[e]) * 5);
    }
    return;
  }
  if (cut_a < g0 || cut_a >= g0 + kThreads) cut_a = -1;
  if (cut_b < g0 || cut_b >= g0 + kThreads) cut_b = -1;
  if (tid < 8 * (kPlanes + 1)) red[tid / (kPlanes + 1)][tid % (kPlanes + 1)] = 0ull;
  const int gidx = g0 + tid;
  const bool active = gidx < ngroups;
  const int c = active ? 8 * gidx : 0;
  // side of the lane's group for either lag (2: its sums are not used) and where its second-strip pixels start
  int off_a = 0, off_b = 0, side_a = 2, side_b = 2;
  if (active && gidx != cut_a) {
    side_a = c >= m ? 1 : 0;
    off_a = c >= m ? c - m : c - m + L;
  }
  if (active && gidx != cut_b) {
    side_b = c + m >= L ? 1 : 0;
    off_b = c + m >= L ? c + m - L : c + m;
  }
  // odd lags (m odd: both at once): a lane's 16-byte loads start one pixel early, on a 4-byte boundary, and its last
  // pixel comes from the lane after it, unless that lane loads something else
  const bool odd = (m & 1) != 0;
  auto lonely_at = [&](int cut) {
    return (tid & 63) == 63 || gidx + 1 >= ngroups || gidx + 1 == cut;
  };
  const bool lonely_a = lonely_at(cut_a), lonely_b = lonely_at(cut_b);
  SweepAcc<4> sa;
  const uint16_t* a_lane = d.a + c;
  // row r of the first strip meets row r - ds of the second
  const uint16_t* b_rows[2] = {d.b + (int64_t)(-ds[0]) * d.pitch, d.b + (int64_t)(-ds[1]) * d.pitch};
#define MS_SWEEP_WITH(LA, LB, ODD)                                                                                     \
  do {                                                                                                                 \
    if (sig_live[0] && sig_live[1])                                                                                    \
      sweep_rows_both<LA, LB, ODD>(a_lane, b_rows[0], b_rows[1], d.pitch, r0, r1, off_a, off_b, lonely_a, lonely_b, sa); \
    else if (sig_live[0])                                                                                              \
      sweep_rows<LA, LB, ODD, 0>(a_lane, b_rows[0], d.pitch, r0, r1, off_a, off_b, lonely_a, lonely_b, sa);            \
    else                                                                                                               \
      sweep_rows<LA, LB, ODD, 2>(a_lane, b_rows[1], d.pitch, r0, r1, off_a, off_b, lonely_a, lonely_b, sa);            \
  } while (0)
#define MS_SWEEP(LA, LB)                 \
  do {                                   \
    if (odd) MS_SWEEP_WITH(LA, LB, true); \
    else MS_SWEEP_WITH(LA, LB, false);   \
  } while (0)
  if (lag_a && lag_b) MS_SWEEP(true, true);
  else if (lag_a) MS_SWEEP(true, false);
  else MS_SWEEP(false, true);
#undef MS_SWEEP
#undef MS_SWEEP_WITH
  __syncthreads();  // red is zero
#pragma unroll
  for (int sig = 0; sig < 2; ++sig) {
    if (!sig_live[sig]) continue;  // uniform
#pragma unroll
    for (int lag = 0; lag < 2; ++lag) {
      if (!(lag ? lag_b : lag_a)) continue;  // uniform
      const int side = lag ? side_b : side_a;
      unsigned v[kPlanes];
      sa.planes(2 * sig + lag, v);
#pragma unroll
      for (int sd = 0; sd < 2; ++sd) {
        if (__ballot(side == sd) == 0ull) continue;  // wave-uniform: no lane of this wave on that side
        add_planes(v, side == sd, red[4 * sig + 2 * lag + sd]);
      }
    }
  }
  // the cut groups, pixel by pixel: wave w takes the pixels of lag w / 2 on side w % 2 (before the wrap: side 0 for
  // either lag), lane l pixel l % 8 of them in rows r0 + l / 8, + 8, ..., for one row lag after the other
  {
    const int w = tid >> 6, lane = tid & 63;
    const int lag = w >> 1, sd = w & 1;
    const int cut = lag ? cut_b : cut_a;
    if (cut >= 0) {  // wave-uniform
      const int shift = lag ? m : -m;
      const int first = 8 * cut;
      const int before = lag ? L - m - first : m - first;  // pixels of the group before the wrap: 1..7
      const int i = (lane & 7) + (sd ? before : 0);
      const bool mine = sd ? i < 8 : i < before;
      int bc = first + i + shift;
      bc += bc < 0 ? L : 0;
      bc -= bc >= L ? L : 0;
#pragma unroll 1
      for (int sig = 0; sig < 2; ++sig) {
        if (!sig_live[sig]) continue;  // uniform
        MomentAcc ma;
        if (mine) {
          for (int r = r0 + (lane >> 3); r < r1; r += 8)
            ma.add1(d.a[(int64_t)r * d.pitch + first + i], b_rows[sig][(int64_t)r * d.pitch + bc]);
        }
        const unsigned v[kPlanes] = {ma.a_l,   ma.a_h,   ma.b_l,   ma.b_h,   ma.aa_ll, ma.aa_lh, ma.aa_hh,
                                     ma.bb_ll, ma.bb_lh, ma.bb_hh, ma.ab_ll, ma.ab_x,  ma.ab_hh};
        add_planes(v, true, red[4 * sig + 2 * lag + sd]);
      }
    }
  }
  __syncthreads();
  if (tid < 40) {
    const int e = tid / 5, i = tid % 5;
    const unsigned long long* pl = red[e];
    unsigned long long v;
    if (i == 0) v = pl[0] + (pl[1] << 8);
    else if (i == 1) v = pl[2] + (pl[3] << 8);
    else if (i == 2) v = pl[4] + (pl[5] << 9) + (pl[6] << 16);
    else if (i == 3) v = pl[7] + (pl[8] << 9) + (pl[9] << 16);
    else v = pl[10] + (pl[11] << 8) + (pl[12] << 16);
    bool alive = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) alive = (k == e) ? live[k] : alive;
    int slot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) slot = (k == e) ? ci[k] : slot;
    if (alive && v != 0ull) atomicAdd(&acc[((int64_t)p * slots + slot) * 5 + i], v);
  }
}

// ---- exact 128-bit helpers -------------------------------------------------
struct U128 {
  unsigned long long hi, lo;
};
__device__ __forceinline__ U128 mul64(unsigned long long a, unsigned long long b) {
  return U128{__umul64hi(a, b), a * b};
}
__device__ __forceinline__ bool ge128(U128 a, U128 b) { return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo); }
__device__ __forceinline__ U128 sub128(U128 a, U128 b) {  // a >= b
  U128 r;
  r.lo = a.lo - b.lo;
  r.hi = a.hi - b.hi - (a.lo < b.lo ? 1ull : 0ull);
  return r;
}
__device__ __forceinline__ double to_double(U128 a) {
  return (double)a.hi * 18446744073709551616.0 + (double)a.lo;
}

// Pearson correlation from exact moments over n samples; mirrors the special
// cases of the reference's ncc(): constant-constant -> 1, constant-varying -> -inf.
__device__ double ncc_from_moments(unsigned long long n, unsigned long long sa, unsigned long long sb,
                                   unsigned long long saa, unsigned long long sbb, unsigned long long sab) {
  const U128 n_ab = mul64(n, sab), a_b = mul64(sa, sb);
  const U128 n_aa = mul64(n, saa), a_a = mul64(sa, sa);
  const U128 n_bb = mul64(n, sbb), b_b = mul64(sb, sb);
  const U128 va = sub128(n_aa, a_a), vb = sub128(n_bb, b_b);  // n * sum x^2 >= (sum x)^2
  const bool za = (va.hi | va.lo) == 0, zb = (vb.hi | vb.lo) == 0;
  if (za || zb) return (za && zb) ? 1.0 : -INFINITY;
  double num;
  if (ge128(n_ab, a_b))
    num = to_double(sub128(n_ab, a_b));
  else
    num = -to_double(sub128(a_b, n_ab));
  const double den = sqrt(to_double(va) * to_double(vb));
  double r = num / den;
  if (!(r == r) || r == INFINITY || r == -INFINITY) return -INFINITY;
  return r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r);
}

// One wave per pair: the lanes score the candidates 64 at a time in evaluation order; the winner is the first
// candidate that no other beats (the serial walk's strict ">").  The subpixel estimate, in the slot behind the
// candidates, is listed BEFORE the two-peak discrete result by the reference and wins ties against it
// (_translation_computation.py:1232-1313), but AFTER the consensus stage's result, which wins ties against it
// (`wide`: the 160-candidate set of that stage is in use, :1197-1230).
// `table` (optional): [n_pairs][n_cand][3] = y, x, ncc of every candidate in evaluation order, strip relative;
// ncc NaN where the reference does not score the candidate (empty overlap).
//
// `consensus` (only with the 160-candidate set): the consensus stage itself as the reference runs it when
// scikit-image's ransac raises on its model class (ransac_translation_estimation, :1085-1127): among the candidates
// of finite NCC, in evaluation order, the first one with the most neighbours within 2 pixels (Euclidean, itself
// included) is that stage's result when those neighbours are at least 0.3 of the candidates; otherwise, and with
// fewer than 3 candidates (:1030-1038), the first candidate of the largest NCC.  The result list is then
// (consensus stage, subpixel estimate, two-peak discrete search) and the first of the largest NCC wins (:1313).
constexpr int kConsensusMinSamples = 3;          // RANSACConfig.min_samples (:74)
constexpr int kConsensusThreshold2 = 4;          // residual_threshold = 2.0, squared (:75; integer shifts: exact)
constexpr double kConsensusMinInlierRatio = 0.3;  // min_inlier_ratio (:78)
constexpr int kConsensusFar = 20000;             // row of a candidate that does not take part: no neighbour of any shift
__global__ __launch_bounds__(256) void k_finalize(const PairDesc* desc, const int8_t* status_in, int n_pairs,
                                                  const Cand* cand, const unsigned long long* acc, double* out,
                                                  int8_t* status, const UpsState* ups, int n_cand, int wide,
                                                  double* table, int consensus) {
  __shared__ int s_y[4][kNumCandWide], s_x[4][kNumCandWide];
  __shared__ double s_v[4][kNumCandWide];
  const int wv = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wv;
  const int lane = threadIdx.x & 63;
  if (p >= n_pairs) return;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  const int slots = n_cand + 1;
  if (status_in[p] != MISTITCH_PAIR_OK) {
    if (lane < 3) out[3 * p + lane] = nan;
    if (table != nullptr)
      for (int e = lane; e < 3 * n_cand; e += 64) table[(int64_t)p * n_cand * 3 + e] = nan;
    return;
  }
  const PairDesc d = desc[p];
  const Cand* my_cand = cand + (int64_t)p * slots;
  const unsigned long long* my_acc = acc + (int64_t)p * slots * 5;
  auto score = [&](const Cand& k, int index) -> double {
    // a duplicate takes the value of the candidate it repeats (always an earlier, scored one: same moments)
    const int src = k.flag == 3 ? k.dup_of : index;
    const unsigned long long* m = my_acc + (int64_t)src * 5;
    const int ay = k.y < 0 ? -k.y : k.y, ax = k.x < 0 ? -k.x : k.x;
    const unsigned long long n = (unsigned long long)(d.img_h - ay) * (unsigned long long)(d.img_w - ax);
    return ncc_from_moments(n, m[0], m[1], m[2], m[3], m[4]);
  };
  double best = -INFINITY;
  int by = 0, bx = 0;
  bool any = false;
  for (int c0 = 0; c0 < n_cand; c0 += 64) {
    const int ci = c0 + lane;
    Cand k{0, 0, 1, -1};
    double v = -INFINITY;
    bool evaluated = false;  // an empty overlap (flag 1) is not evaluated at all
    if (ci < n_cand) {
      k = my_cand[ci];
      evaluated = k.flag != 1;
      if (k.flag == 0 || k.flag == 3) v = score(k, ci);
      if (table != nullptr) {
        double* row = table + ((int64_t)p * n_cand + ci) * 3;
        row[0] = (double)k.y;
        row[1] = (double)k.x;
        row[2] = evaluated ? v : nan;
      }
      if (consensus) {  // only candidates of finite NCC reach the consensus stage (:985-986)
        const bool finite = evaluated && v != -INFINITY;
        s_y[wv][ci] = finite ? k.y : kConsensusFar;
        s_x[wv][ci] = finite ? k.x : 0;
        s_v[wv][ci] = finite ? v : -INFINITY;
      }
    }
    double chunk_best = evaluated ? v : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) chunk_best = fmax(chunk_best, __shfl_xor(chunk_best, off));
    const unsigned long long at_best = __ballot(evaluated && v == chunk_best && chunk_best != -INFINITY);
    if (at_best && chunk_best > best) {  // strictly: an equal value later in the walk does not replace
      const int win = __builtin_ctzll(at_best);
      best = chunk_best;
      by = __shfl(k.y, win);
      bx = __shfl(k.x, win);
      any = true;
    }
  }
  double oy = any ? (double)by : 0.0, ox = any ? (double)bx : 0.0;
  double m3 = -INFINITY;  // consensus mode: the two-peak discrete search, listed last
  int m3y = 0, m3x = 0;
  if (consensus) {
    __builtin_amdgcn_wave_barrier();  // the wave's own LDS writes above are read below, by other lanes
    int n_finite = 0, key_best = -1;
    for (int c0 = 0; c0 < n_cand; c0 += 64) {
      const int ci = c0 + lane;
      int count = 0;
      bool finite = false;
      if (ci < n_cand) {
        const int y = s_y[wv][ci], x = s_x[wv][ci];
        finite = y != kConsensusFar;
        if (finite) {
          for (int j = 0; j < n_cand; ++j) {
            const int dy = y - s_y[wv][j], dx = x - s_x[wv][j];
            count += (dy * dy + dx * dx <= kConsensusThreshold2) ? 1 : 0;
          }
        }
      }
      n_finite += __popcll(__ballot(finite));
      // most neighbours first, then the earliest candidate (numpy.argmax)
      key_best = max(key_best, finite ? count * 256 + (255 - ci) : -1);
    }
    static_assert(kNumCandWide <= 256, "candidate index packed into 8 bits");
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) key_best = max(key_best, __shfl_xor(key_best, off));
    if (n_finite >= kConsensusMinSamples) {
      const int count = key_best >> 8, idx = 255 - (key_best & 255);
      if (!((double)count / (double)n_finite < kConsensusMinInlierRatio)) {  // the consensus stands
        best = s_v[wv][idx];
        oy = (double)s_y[wv][idx];
        ox = (double)s_x[wv][idx];
      }
    }
    const double mine = lane < kNumCand ? s_v[wv][lane] : -INFINITY;
    double top = mine;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) top = fmax(top, __shfl_xor(top, off));
    const unsigned long long at_top = __ballot(mine == top && top != -INFINITY);
    if (at_top) {
      const int win = __builtin_ctzll(at_top);
      m3 = top;
      m3y = s_y[wv][win];
      m3x = s_x[wv][win];
    }
  }
  const bool has_ups = ups != nullptr && ups[p].valid;
  if (has_ups) {
    const Cand ku = my_cand[n_cand];
    if (ku.flag == 0) {
      const double v_ups = score(ku, n_cand);
      if (v_ups != -INFINITY && (wide ? v_ups > best : v_ups >= best)) {
        best = v_ups;
        oy = d.transposed ? ups[p].sub_l : ups[p].sub_s;
        ox = d.transposed ? ups[p].sub_s : ups[p].sub_l;
      }
    }
  }
  if (m3 > best) {  // (consensus mode) the discrete search only wins with a strictly larger NCC: it is listed last
    best = m3;
    oy = (double)m3y;
    ox = (double)m3x;
  }
  if (lane == 0) {
    out[3 * p] = best;
    out[3 * p + 1] = oy + (double)d.off_y;
    out[3 * p + 2] = ox + (double)d.off_x;
    if (best == -INFINITY) status[p] = MISTITCH_PAIR_NO_CANDIDATE;
  }
}

// NCC of the overlap of one pair's strips under the strip-relative shift (ty, tx), computed
// by the whole workgroup; every thread returns the value.  -inf for an empty overlap or fewer
// than 25 pixels, as extract_overlap_subregion + ncc give (_translation_computation.py:535-657).
__device__ double block_ncc(const PairDesc& d, int ty, int tx, unsigned long long* red /*[5 * kThreads / 64 + 1]*/) {
  const int ay = ty < 0 ? -ty : ty, ax = tx < 0 ? -tx : tx;
  if (ay >= d.img_h || ax >= d.img_w) return -INFINITY;
  const unsigned long long n = (unsigned long long)(d.img_h - ay) * (unsigned long long)(d.img_w - ax);
  if (n < 25) return -INFINITY;
  const int ds = d.transposed ? tx : ty, dl = d.transposed ? ty : tx;
  const int s_lo = ds > 0 ? ds : 0, s_hi = d.S + (ds < 0 ? ds : 0);
  const int l_lo = dl > 0 ? dl : 0, l_hi = d.L + (dl < 0 ? dl : 0);
  unsigned long long v[5];
  moments_of_rows<true>(d.a + l_lo, d.pitch, d.b + (int64_t)(-ds) * d.pitch + (l_lo - dl), d.pitch, s_lo, s_hi, 1,
                        l_hi - l_lo, v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) red[wave * 5 + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < kThreads / 64; ++w)
      for (int i = 0; i < 5; ++i) t[i] += red[w * 5 + i];
    const double r = ncc_from_moments(n, t[0], t[1], t[2], t[3], t[4]);
    red[5 * kThreads / 64] = (unsigned long long)__double_as_longlong(r);
  }
  __syncthreads();
  return __longlong_as_double((long long)red[5 * kThreads / 64]);
}

// Integer hill climb of the strip NCC around a start shift, one workgroup per pair:
// find_local_max_integer_constrained driven as refine_translations drives it
// (registration/_constrained_refinement.py:62-138, 229-295).  start and the result are
// shifts of the second tile in the first tile's frame; row (NaN, start) where the reference
// keeps the start (start outside its limits, start not scoring).
__global__ __launch_bounds__(kThreads) void k_refine_pairs(const PairDesc* desc, const int8_t* status, int pair0,
                                                           const int32_t* __restrict__ start, double r, int H, int W,
                                                           double* out) {
  __shared__ unsigned long long red[5 * kThreads / 64 + 1];
  const int p = pair0 + blockIdx.x;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  const int iy = start[2 * p], ix = start[2 * p + 1];
  if (status[p] != MISTITCH_PAIR_OK) {
    if (threadIdx.x == 0) {
      out[3 * p] = nan;
      out[3 * p + 1] = (double)iy;
      out[3 * p + 2] = (double)ix;
    }
    return;
  }
  const PairDesc d = desc[p];
  const double lo_y = fmax((double)(-H + 1), (double)iy - r), hi_y = fmin((double)(H - 1), (double)iy + r);
  const double lo_x = fmax((double)(-W + 1), (double)ix - r), hi_x = fmin((double)(W - 1), (double)ix + r);
  int y = iy, x = ix;
  double value = nan;
  bool ok = lo_y <= (double)iy && (double)iy <= hi_y && lo_x <= (double)ix && (double)ix <= hi_x;
  if (ok) {
    value = block_ncc(d, y - d.off_y, x - d.off_x, red);
    ok = value != INFINITY && value != -INFINITY && value == value;
    if (!ok) value = nan;
  }
  if (ok) {
    double prev = -INFINITY;
    for (int it = 0; it < 100; ++it) {
      double best = -INFINITY;
      int by = 0, bx = 0;
      bool any = false;
      for (int dy = -1; dy <= 1; ++dy) {
        for (int dx = -1; dx <= 1; ++dx) {
          const double ny = (double)(y + dy), nx = (double)(x + dx);
          if (!(lo_y <= ny && ny <= hi_y && lo_x <= nx && nx <= hi_x)) continue;
          const double v = block_ncc(d, y + dy - d.off_y, x + dx - d.off_x, red);
          if (!any || v > best) {  // first of equal maxima, as numpy.argmax
            best = v;
            by = y + dy;
            bx = x + dx;
            any = true;
          }
        }
      }
      if (!any) break;
      if (best <= value || fabs(best - prev) < 1e-6) break;
      y = by;
      x = bx;
      prev = value;
      value = best;
    }
  }
  if (threadIdx.x == 0) {
    out[3 * p] = value;
    out[3 * p + 1] = (double)y;
    out[3 * p + 2] = (double)x;
  }
}

__global__ __launch_bounds__(kThreads) void k_ncc_single(const uint16_t* a, int64_t pa, const uint16_t* b, int64_t pb,
                                                         int h, int w, unsigned long long* acc) {
  __shared__ unsigned long long red[5 * kThreads / 64];
  unsigned long long v[5];
  moments_of_rows<true>(a, pa, b, pb, blockIdx.x, h, gridDim.x, w, v);
  reduce_moments(v, red, acc);
}

__global__ void k_ncc_single_finish(const unsigned long long* acc, long long n, long long min_n, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    out[0] = (n == 0 || n < min_n) ? -INFINITY
                                   : ncc_from_moments((unsigned long long)n, acc[0], acc[1], acc[2], acc[3], acc[4]);
}

__global__ void k_peaks_to_arrays(const Peak* peaks, int k, int img_w, int32_t* rows, int32_t* cols,
                                  double* vals) {
  const int i = threadIdx.x;
  if (i >= k) return;
  const Peak pk = peaks[i];
  rows[i] = pk.key / img_w;
  cols[i] = pk.key - (pk.key / img_w) * img_w;
  vals[i] = pk.val;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// The passes of a length with odd factors: its odd prime factors (each at most 13) in descending
// order, then the power of two that is left in radices of at most 8.  False when a larger prime
// divides n or the passes would be more than six.
static bool mixed_radices(int n, std::vector<int>* out) {
  out->clear();
  for (int r : {13, 11, 7, 5, 3})
    while (n % r == 0) {
      out->push_back(r);
      n /= r;
    }
  if (n & (n - 1)) return false;
  int lg = 0;
  while ((1 << lg) < n) ++lg;
  if (lg % 3) out->push_back(1 << (lg % 3));
  for (int i = 0; i < lg / 3; ++i) out->push_back(8);
  return !out->empty() && out->size() <= 6;
}

static int build_plan(mistitch_ctx* ctx, int n, bool f64, FftPlanDev* out, int max_radix = 16) {
  const long long key = ((long long)n * 2 + (f64 ? 1 : 0)) * 32 + max_radix;
  auto it = ctx->plans.find(key);
  if (it != ctx->plans.end()) {
    *out = it->second.dev;
    return MISTITCH_OK;
  }
  FftPlanHost ph;
  ph.dev.n = n;
  // remainder radix first, then 16s: the last (largest) radix sets how many
  // adjacent storage positions share their low frequency digits.
  int np = 0;
  if (is_pow2(n)) {
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    const int lr = max_radix >= 16 ? 4 : (max_radix >= 8 ? 3 : 2);
    if (lg % lr) ph.dev.radix[np++] = 1 << (lg % lr);
    for (int i = 0; i < lg / lr; ++i) ph.dev.radix[np++] = 1 << lr;
  } else {
    std::vector<int> radices;
    MS_REQUIRE(mixed_radices(n, &radices), MISTITCH_ERR_UNSUPPORTED, "cannot factor FFT length %d into radices up to 13", n);
    for (int r : radices) ph.dev.radix[np++] = r;
  }
  ph.dev.npass = np;
  {
    long long prod = 1;
    for (int i = 0; i < np; ++i) prod *= ph.dev.radix[i];
    MS_REQUIRE(prod == n && np <= 6, MISTITCH_ERR_UNSUPPORTED, "cannot factor FFT length %d", n);
  }
  std::vector<uint16_t> pos2freq(n), freq2pos(n), mpos(n);
  for (int pos = 0; pos < n; ++pos) {
    int rest = pos, span = n, mult = 1, k = 0;
    for (int i = 0; i < np; ++i) {
      span /= ph.dev.radix[i];
      const int digit = rest / span;
      rest -= digit * span;
      k += digit * mult;
      mult *= ph.dev.radix[i];
    }
    pos2freq[pos] = (uint16_t)k;
    freq2pos[k] = (uint16_t)pos;
  }
  for (int pos = 0; pos < n; ++pos) mpos[pos] = freq2pos[(n - pos2freq[pos]) % n];
  const size_t esz = f64 ? 16 : 8;
  const size_t tw_bytes = (size_t)n * esz;
  const size_t tab_bytes = (size_t)n * sizeof(uint16_t);
  const size_t tab_stride = (tab_bytes + 255) / 256 * 256;
  const size_t total = (tw_bytes + 255) / 256 * 256 + 3 * tab_stride;
  std::vector<char> host(total, 0);
  for (int k = 0; k < n; ++k) {
    const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
    const long double c = cosl(ang), s = sinl(ang);
    if (f64) {
      double* t = reinterpret_cast<double*>(host.data()) + 2 * k;
      t[0] = (double)c;
      t[1] = (double)s;
    } else {
      float* t = reinterpret_cast<float*>(host.data()) + 2 * k;
      t[0] = (float)c;
      t[1] = (float)s;
    }
  }
  const size_t off0 = (tw_bytes + 255) / 256 * 256;
  memcpy(host.data() + off0, pos2freq.data(), tab_bytes);
  memcpy(host.data() + off0 + tab_stride, freq2pos.data(), tab_bytes);
  memcpy(host.data() + off0 + 2 * tab_stride, mpos.data(), tab_bytes);
  MS_HIP(hipMalloc(&ph.blob, total));
  MS_HIP(hipMemcpy(ph.blob, host.data(), total, hipMemcpyHostToDevice));
  char* base = reinterpret_cast<char*>(ph.blob);
  ph.dev.tw = base;
  ph.dev.pos2freq = reinterpret_cast<const uint16_t*>(base + off0);
  ph.dev.freq2pos = reinterpret_cast<const uint16_t*>(base + off0 + tab_stride);
  ph.dev.mpos = reinterpret_cast<const uint16_t*>(base + off0 + 2 * tab_stride);
  ctx->plans[key] = ph;
  *out = ph.dev;
  return MISTITCH_OK;
}

// Cut the storage positions of the long axis into column-pass units of (at most)
// 2C columns: aligned blocks of C positions paired with the block holding their
// negated frequencies, or merged into one self-contained range where negation
// does not map block to block (the positions of the frequencies 0, M, 2M, ...).
static bool build_units(const std::vector<uint16_t>& mpos, int L, int C, std::vector<CpsUnit>* units,
                        std::vector<uint16_t>* stored_pos, std::vector<uint16_t>* mirror_fill) {
  const int nb = L / C;
  if (nb < 1) return false;
  std::vector<char> done(nb, 0);
  units->clear();
  for (int b = 0; b < nb; ++b) {
    if (done[b]) continue;
    std::set<int> group{b};
    bool grew = true;
    while (grew) {
      grew = false;
      for (int g : std::vector<int>(group.begin(), group.end()))
        for (int c = 0; c < C; ++c)
          if (group.insert(mpos[g * C + c] / C).second) grew = true;
    }
    const int lo = *group.begin(), hi = *group.rbegin();
    if (group.size() == 1) {
      units->push_back(CpsUnit{b * C, C, 0, 0});
    } else if (group.size() == 2) {
      // two blocks: either each maps wholly onto the other (a pair) or they only close together
      bool pair = true;
      for (int c = 0; c < C; ++c) pair = pair && (mpos[lo * C + c] / C == hi);
      if (pair)
        units->push_back(CpsUnit{lo * C, C, hi * C, C});
      else if (hi == lo + 1)
        units->push_back(CpsUnit{lo * C, 2 * C, 0, 0});
      else
        return false;
    } else {
      return false;
    }
    for (int g : group) done[g] = 1;
  }
  // what the final long-axis pass reads: the stored positions, and for each the
  // position (if any) that only exists as its conjugate mirror
  std::vector<char> stored(L, 0);
  for (const CpsUnit& u : *units)
    for (int i = 0; i < u.na; ++i) stored[u.a0 + i] = 1;
  stored_pos->clear();
  mirror_fill->clear();
  std::vector<char> covered(L, 0);
  for (int pos = 0; pos < L; ++pos) {
    if (!stored[pos]) continue;
    stored_pos->push_back((uint16_t)pos);
    covered[pos] = 1;
    const int m = mpos[pos];
    if (!stored[m]) {
      mirror_fill->push_back((uint16_t)m);
      covered[m] = 1;
    } else {
      mirror_fill->push_back(0xffff);
    }
  }
  for (int pos = 0; pos < L; ++pos)
    if (!covered[pos]) return false;
  return true;
}

struct ShapeCfg {
  int S, L;
  bool gen_short;   // the short side is not a power of two (complex128 only)
  bool mixed_short; // ... but a product of radices up to 13: mixed-radix passes on planS; chirp-z on gpS otherwise
  GenPlanDev gpS;   // the chirp-z plan of the short axis then (planS unused)
  FftPlanDev planS, planL;
  int C;       // columns per block of the column pass (LDS rows hold 2C)
  bool reg_cols;  // every unit is 8 stored columns, alone or with the 8 of their negated frequencies (k_fft_short_cps_reg)
  const CpsUnit* units_paired;  // device: the units with a partner block, and those without (their own mirrors)
  const CpsUnit* units_single;
  int n_paired, n_single;
  const CpsUnit* units;    // device
  int n_units;
  const uint16_t* stored_pos;   // device: positions k_fft_short_cps wrote, ascending
  const uint16_t* mirror_fill;  // device: per stored position, the position that is its conjugate (0xffff: none)
  int n_stored;
  const uint16_t* pos2stored;  // device: position -> index in stored_pos (0xffff: not stored)
  const KappaEntry* kappa;     // device: the stored column of every |frequency| 0 .. L/2 (subpixel stage)
  const uint16_t* stored2kappa;  // device: stored column -> its |frequency| (0xffff: not the column that represents it)
  const uint16_t* ups_slabs;     // device: first position of every aligned group of 8 positions with a stored column
  int n_ups_slabs;
  int NF_fwd;  // lines per k_fft_long_fwd block
  int NF_inv;  // line pairs per k_fft_long_inv block
  size_t lds_fwd, lds_cps, lds_inv;
};

constexpr size_t kLdsBudget = 64 * 1024;  // two blocks per CU
constexpr size_t kLdsMax = 159 * 1024;  // dynamic LDS a kernel may ask for: 160 KiB less room for its static arrays

template <typename T>
static size_t line_bytes(int L) {
  return (size_t)(LineAddr<T>::padded(L) + 1) * sizeof(cx<T>);
}

static int build_gen_plan(mistitch_ctx* ctx, int n, GenPlanDev* out, int max_radix = 16);

static int make_shape(mistitch_ctx* ctx, int S, int L, bool f64, ShapeCfg* cfg) {
  const bool gen_short = !is_pow2(S);
  MS_REQUIRE(is_pow2(L) && L >= 8 && L <= 4096 && S <= 1024 && (gen_short ? (S >= 3 && f64) : S >= 8),
             MISTITCH_ERR_UNSUPPORTED,
             "strip %d x %d unsupported here: the long side must be a power of two in [8, 4096], the short side at most 1024",
             S, L);
  const long long key = (((long long)S << 20) | (long long)L) * 2 + (f64 ? 1 : 0);
  auto it = ctx->shapes.find(key);
  if (it != ctx->shapes.end()) {
    memcpy(cfg, it->second.data(), sizeof(ShapeCfg));
    return MISTITCH_OK;
  }
  cfg->S = S;
  cfg->L = L;
  cfg->gen_short = gen_short;
  std::vector<int> short_radices;
  const bool mixed = gen_short && mixed_radices(S, &short_radices);
  cfg->mixed_short = mixed;
  int rc;
  if (mixed) {
    rc = build_plan(ctx, S, true, &cfg->planS);
    memset(&cfg->gpS, 0, sizeof(cfg->gpS));
  } else if (gen_short) {
    rc = build_gen_plan(ctx, S, &cfg->gpS, kGenShortRadix);
    memset(&cfg->planS, 0, sizeof(cfg->planS));
  } else {
    rc = build_plan(ctx, S, f64, &cfg->planS, f64 ? kShortMaxRadix<double> : kShortMaxRadix<float>);
    memset(&cfg->gpS, 0, sizeof(cfg->gpS));
  }
  if (rc) return rc;
  rc = build_plan(ctx, L, f64, &cfg->planL, f64 ? kLongMaxRadix<double> : kLongMaxRadix<float>);
  if (rc) return rc;
  const size_t esz = f64 ? 16 : 8;
  std::vector<uint16_t> mpos(L);
  MS_HIP(hipMemcpy(mpos.data(), cfg->planL.mpos, (size_t)L * 2, hipMemcpyDeviceToHost));
  // widest column block whose units exist and whose working set leaves room for two blocks per CU
  std::vector<CpsUnit> units;
  std::vector<uint16_t> stored_pos, mirror_fill;
  // (chirp-z columns are lines of M >= 2S - 1 elements: 76 KiB still leaves room for two blocks per CU)
  const bool chirp = gen_short && !mixed;
  const size_t col_bytes = chirp ? line_bytes<double>(cfg->gpS.M) : (size_t)S * esz;
  const size_t col_budget = chirp ? 76 * 1024 : kLdsBudget;
  int C = chirp ? 8 : 16;
  for (; C >= 1; C >>= 1) {
    if (C > L) continue;
    if (C > 1 && col_bytes * 2 * C > col_budget) continue;
    if (build_units(mpos, L, C, &units, &stored_pos, &mirror_fill)) break;
  }
  MS_REQUIRE(C >= 1, MISTITCH_ERR_UNSUPPORTED, "no column-pass decomposition for long side %d", L);
  cfg->C = C;
  cfg->reg_cols = C == 8;
  for (const CpsUnit& u : units) cfg->reg_cols = cfg->reg_cols && u.na == 8 && (u.nb == 8 || u.nb == 0);
  if (mixed)
    cfg->lds_cps = ((size_t)S * 2 * C + S) * esz + ((size_t)S * 4 + 15) / 16 * 16;  // columns, all S roots, two index tables
  else if (gen_short)
    cfg->lds_cps = col_bytes * 2 * C + (size_t)Twiddles<double>::entries(cfg->gpS.M) * 16;
  else
    cfg->lds_cps = ((size_t)S * 2 * C + (f64 ? Twiddles<double>::entries(S) : Twiddles<float>::entries(S))) * esz +
                   ((size_t)S * 4 + 15) / 16 * 16;  // + two uint16 index tables of the short axis
  MS_REQUIRE(cfg->lds_cps <= kLdsMax, MISTITCH_ERR_UNSUPPORTED, "short side %d needs %zu B of LDS", S, cfg->lds_cps);
  {
    std::vector<uint16_t> pos2freq(L), pos2stored(L, 0xffff);
    MS_HIP(hipMemcpy(pos2freq.data(), cfg->planL.pos2freq, (size_t)L * 2, hipMemcpyDeviceToHost));
    std::vector<KappaEntry> kappa(L / 2 + 1, KappaEntry{0xffff, 1, 0});
    for (size_t i = 0; i < stored_pos.size(); ++i) {
      const int pos = stored_pos[i];
      pos2stored[pos] = (uint16_t)i;
      const int f = pos2freq[pos];
      const int k = f < L / 2 ? f : f - L;  // Nyquist counts as negative
      const int a = k < 0 ? -k : k;
      if (kappa[a].stored != 0xffff) continue;  // both columns of the pair are stored: one represents it
      kappa[a] = KappaEntry{(uint16_t)i, (int8_t)(k < 0 ? -1 : 1), (int8_t)(mpos[pos] == pos ? 1 : 0)};
    }
    for (const KappaEntry& k : kappa)
      MS_REQUIRE(k.stored != 0xffff, MISTITCH_ERR_UNSUPPORTED, "a conjugate pair of columns is not stored (L = %d)", L);
    const size_t ub = units.size() * sizeof(CpsUnit), sb = stored_pos.size() * sizeof(uint16_t);
    const size_t ub_pad = (ub + 15) / 16 * 16, sb_pad = (sb + 15) / 16 * 16, pb = ((size_t)L * 2 + 15) / 16 * 16;
    const size_t kb = (kappa.size() * sizeof(KappaEntry) + 15) / 16 * 16;
    std::vector<uint16_t> s2k(stored_pos.size(), 0xffff);
    for (size_t a = 0; a < kappa.size(); ++a) s2k[kappa[a].stored] = (uint16_t)a;
    // the subpixel stage walks every aligned group of 8 positions that holds a stored column (columns of a
    // group that are not stored are computed along and dropped)
    std::vector<uint16_t> ups_slabs;
    {
      std::vector<char> has(L / kSlab, 0);
      for (uint16_t pos : stored_pos) has[pos / kSlab] = 1;
      for (int i = 0; i < L / kSlab; ++i)
        if (has[i]) ups_slabs.push_back((uint16_t)(i * kSlab));
    }
    void* blob = nullptr;
    MS_HIP(hipMalloc(&blob, ub_pad + 2 * sb_pad + pb + kb + 3 * sb_pad + 64 + ub_pad + 16));
    ctx->shape_allocs.push_back(blob);
    char* base = reinterpret_cast<char*>(blob);
    MS_HIP(hipMemcpy(base, units.data(), ub, hipMemcpyHostToDevice));
    {
      // the same units, those with a partner block first (k_fft_short_cps_reg takes the two kinds in separate launches)
      std::vector<CpsUnit> sorted_units;
      for (const CpsUnit& u : units) if (u.nb != 0) sorted_units.push_back(u);
      cfg->n_paired = (int)sorted_units.size();
      for (const CpsUnit& u : units) if (u.nb == 0) sorted_units.push_back(u);
      cfg->n_single = (int)sorted_units.size() - cfg->n_paired;
      char* at = base + ub_pad + 2 * sb_pad + pb + kb + 3 * sb_pad + 64;
      at = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(at) + 15) & ~(uintptr_t)15);
      MS_HIP(hipMemcpy(at, sorted_units.data(), ub, hipMemcpyHostToDevice));
      cfg->units_paired = reinterpret_cast<const CpsUnit*>(at);
      cfg->units_single = cfg->units_paired + cfg->n_paired;
    }
    MS_HIP(hipMemcpy(base + ub_pad, stored_pos.data(), sb, hipMemcpyHostToDevice));
    MS_HIP(hipMemcpy(base + ub_pad + sb_pad, mirror_fill.data(), sb, hipMemcpyHostToDevice));
    MS_HIP(hipMemcpy(base + ub_pad + 2 * sb_pad, pos2stored.data(), (size_t)L * 2, hipMemcpyHostToDevice));
    MS_HIP(hipMemcpy(base + ub_pad + 2 * sb_pad + pb, kappa.data(), kappa.size() * sizeof(KappaEntry), hipMemcpyHostToDevice));
    MS_HIP(hipMemcpy(base + ub_pad + 2 * sb_pad + pb + kb, s2k.data(), sb, hipMemcpyHostToDevice));
    cfg->units = reinterpret_cast<const CpsUnit*>(base);
    cfg->n_units = (int)units.size();
    cfg->stored_pos = reinterpret_cast<const uint16_t*>(base + ub_pad);
    cfg->mirror_fill = reinterpret_cast<const uint16_t*>(base + ub_pad + sb_pad);
    cfg->n_stored = (int)stored_pos.size();
    cfg->pos2stored = reinterpret_cast<const uint16_t*>(base + ub_pad + 2 * sb_pad);
    cfg->kappa = reinterpret_cast<const KappaEntry*>(base + ub_pad + 2 * sb_pad + pb);
    cfg->stored2kappa = reinterpret_cast<const uint16_t*>(base + ub_pad + 2 * sb_pad + pb + kb);
    char* slabs_at = base + ub_pad + 2 * sb_pad + pb + kb + sb_pad;
    if (!ups_slabs.empty()) MS_HIP(hipMemcpy(slabs_at, ups_slabs.data(), ups_slabs.size() * 2, hipMemcpyHostToDevice));
    cfg->ups_slabs = reinterpret_cast<const uint16_t*>(slabs_at);
    cfg->n_ups_slabs = (int)ups_slabs.size();
  }
  const size_t lb = f64 ? line_bytes<double>(L) : line_bytes<float>(L);
  int nf = (int)(kLdsBudget / lb);
  if (nf < 1) nf = 1;
  int want = kThreads * 16 / L;  // enough butterflies to occupy every lane
  if (want < 1) want = 1;
  if (nf > want) nf = want;
  while (nf & (nf - 1)) nf &= nf - 1;  // round down to a power of two
  cfg->NF_fwd = nf;  // a power of two that divides S: whole blocks of lines
  while (cfg->NF_fwd > S || S % cfg->NF_fwd) cfg->NF_fwd >>= 1;
  cfg->NF_inv = nf > (S + 1) / 2 ? (S + 1) / 2 : nf;
  while (cfg->NF_inv & (cfg->NF_inv - 1)) cfg->NF_inv &= cfg->NF_inv - 1;
  const size_t twb = (size_t)(f64 ? Twiddles<double>::entries(L) : Twiddles<float>::entries(L)) * esz;
  cfg->lds_fwd = lb * cfg->NF_fwd + twb;
  cfg->lds_inv = lb * cfg->NF_inv + twb;
  MS_REQUIRE(cfg->lds_fwd <= kLdsMax, MISTITCH_ERR_UNSUPPORTED, "long side %d needs %zu B of LDS", L, cfg->lds_fwd);
  std::vector<char> blob(sizeof(ShapeCfg));
  memcpy(blob.data(), cfg, sizeof(ShapeCfg));
  ctx->shapes[key] = blob;
  return MISTITCH_OK;
}

template <typename KernelT>
static int allow_lds(mistitch_ctx* ctx, KernelT kernel, size_t bytes) {
  const void* fn = reinterpret_cast<const void*>(kernel);
  if (bytes > 48 * 1024 && !ctx->lds_enabled.count(fn)) {
    MS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax));
    ctx->lds_enabled.insert(fn);
  }
  return MISTITCH_OK;
}

// ---- strips with a side that is not a power of two ----------------------------
struct GenShape {
  int S, L;
  GenPlanDev ps, pl;  // short axis, long axis
  int C;              // columns per k_gen_cols block
  size_t lds_rows, lds_cols;
  // subpixel stage: the columns of frequency 0 .. L/2 are "stored" in natural order
  int Sr;             // rows of its slabs: S, or S + 1 when S is odd
  int n_cols8;        // L/2 + 1 rounded up to whole slabs
  const uint16_t* ups_slabs;     // device
  const uint16_t* ups_identity;  // device: position -> stored column, and stored column -> |frequency| (0xffff past L/2)
  const uint16_t* ups_pos;       // device: position -> stored column (identity throughout)
  const KappaEntry* kappa;       // device
};

// In-place radix-2 FFT in long double (host, plan construction only).
static void host_fft(std::vector<long double>& re, std::vector<long double>& im) {
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(re[i], re[j]);
      std::swap(im[i], im[j]);
    }
  }
  const long double pi = 3.14159265358979323846264338327950288L;
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t k = 0; k < len / 2; ++k) {
      const long double ang = -2.0L * pi * (long double)k / (long double)len;
      const long double wr = cosl(ang), wi = sinl(ang);
      for (size_t i = k; i < n; i += len) {
        const size_t j = i + len / 2;
        const long double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
        re[j] = re[i] - tr;
        im[j] = im[i] - ti;
        re[i] += tr;
        im[i] += ti;
      }
    }
  }
}

static int build_gen_plan(mistitch_ctx* ctx, int n, GenPlanDev* out, int max_radix) {
  const int plan_key = n * 32 + max_radix;
  auto it = ctx->gen_plans.find(plan_key);
  if (it != ctx->gen_plans.end()) {
    memcpy(out, it->second.data(), sizeof(GenPlanDev));
    return MISTITCH_OK;
  }
  int M = 16;
  while (M < 2 * n - 1) M <<= 1;
  MS_REQUIRE(M <= 8192, MISTITCH_ERR_UNSUPPORTED, "side %d too long for the chirp-z path (at most 4096)", n);
  GenPlanDev gp;
  gp.n = n;
  gp.M = M;
  int rc = build_plan(ctx, M, true, &gp.planM, max_radix);
  if (rc) return rc;
  const long double pi = 3.14159265358979323846264338327950288L;
  std::vector<double> chirp(2 * (size_t)n);
  std::vector<long double> bre(M, 0.0L), bim(M, 0.0L);
  for (int j = 0; j < n; ++j) {
    const long long r = ((long long)j * j) % (2LL * n);  // exp(-i pi j^2 / n) has period 2n in j^2
    const long double ang = -pi * (long double)r / (long double)n;
    const long double c = cosl(ang), sn = sinl(ang);
    chirp[2 * j] = (double)c;
    chirp[2 * j + 1] = (double)sn;
    bre[j] = c;
    bim[j] = -sn;
    if (j > 0) {
      bre[M - j] = c;
      bim[M - j] = -sn;
    }
  }
  host_fft(bre, bim);
  std::vector<uint16_t> pos2freq(M);
  MS_HIP(hipMemcpy(pos2freq.data(), gp.planM.pos2freq, (size_t)M * 2, hipMemcpyDeviceToHost));
  std::vector<double> bhat(2 * (size_t)M);
  for (int pos = 0; pos < M; ++pos) {
    bhat[2 * pos] = (double)(bre[pos2freq[pos]] / (long double)M);
    bhat[2 * pos + 1] = (double)(bim[pos2freq[pos]] / (long double)M);
  }
  void* blob = nullptr;
  const size_t cb = chirp.size() * sizeof(double), bb = bhat.size() * sizeof(double);
  MS_HIP(hipMalloc(&blob, cb + bb));
  ctx->shape_allocs.push_back(blob);
  MS_HIP(hipMemcpy(blob, chirp.data(), cb, hipMemcpyHostToDevice));
  MS_HIP(hipMemcpy(reinterpret_cast<char*>(blob) + cb, bhat.data(), bb, hipMemcpyHostToDevice));
  gp.chirp = reinterpret_cast<const double2*>(blob);
  gp.bhat = reinterpret_cast<const double2*>(reinterpret_cast<char*>(blob) + cb);
  std::vector<char> rec(sizeof(GenPlanDev));
  memcpy(rec.data(), &gp, sizeof(GenPlanDev));
  ctx->gen_plans[plan_key] = rec;
  *out = gp;
  return MISTITCH_OK;
}

static int make_gen_shape(mistitch_ctx* ctx, int S, int L, GenShape* g) {
  MS_REQUIRE(S >= 1 && L >= 1 && S <= 4096 && L <= 4096, MISTITCH_ERR_UNSUPPORTED,
             "strip %d x %d unsupported: sides must be within [1, 4096]", S, L);
  g->S = S;
  g->L = L;
  int rc;
  if ((rc = build_gen_plan(ctx, S, &g->ps))) return rc;
  if ((rc = build_gen_plan(ctx, L, &g->pl))) return rc;
  const size_t line_s = line_bytes<double>(g->ps.M), line_l = line_bytes<double>(g->pl.M);
  const size_t tw_s = (size_t)Twiddles<double>::entries(g->ps.M) * 16, tw_l = (size_t)Twiddles<double>::entries(g->pl.M) * 16;
  g->lds_rows = line_l + tw_l;
  int C = (int)((kLdsBudget - tw_s) / line_s);
  if (C < 1) C = 1;
  if (C > 8) C = 8;
  g->C = C;
  g->lds_cols = line_s * C + tw_s;
  MS_REQUIRE(g->lds_rows <= kLdsMax && g->lds_cols <= kLdsMax, MISTITCH_ERR_UNSUPPORTED,
             "strip %d x %d needs more LDS than a workgroup has", S, L);
  {
    const long long key = ((long long)S << 20) | (long long)L;
    auto it = ctx->gen_tables.find(key);
    if (it == ctx->gen_tables.end()) {
      const int n_cols = L / 2 + 1, n_cols8 = (n_cols + kSlab - 1) / kSlab * kSlab;
      std::vector<uint16_t> slabs(n_cols8 / kSlab), ident(n_cols8), s2k(n_cols8, 0xffff);
      std::vector<KappaEntry> kap(n_cols);
      for (int i = 0; i < n_cols8 / kSlab; ++i) slabs[i] = (uint16_t)(i * kSlab);
      for (int c = 0; c < n_cols8; ++c) ident[c] = (uint16_t)c;
      for (int c = 0; c < n_cols; ++c) {
        s2k[c] = (uint16_t)c;
        const bool nyquist = L % 2 == 0 && c == L / 2;  // counts as the negative frequency -L / 2, as numpy.fft.fftfreq does
        kap[c] = KappaEntry{(uint16_t)c, (int8_t)(nyquist ? -1 : 1), (int8_t)((c == 0 || nyquist) ? 1 : 0)};
      }
      const size_t b1 = (slabs.size() * 2 + 15) / 16 * 16, b2 = ((size_t)n_cols8 * 2 + 15) / 16 * 16;
      const size_t b4 = kap.size() * sizeof(KappaEntry);
      void* blob = nullptr;
      MS_HIP(hipMalloc(&blob, b1 + 2 * b2 + b4));
      ctx->shape_allocs.push_back(blob);
      char* base = reinterpret_cast<char*>(blob);
      MS_HIP(hipMemcpy(base, slabs.data(), slabs.size() * 2, hipMemcpyHostToDevice));
      MS_HIP(hipMemcpy(base + b1, ident.data(), (size_t)n_cols8 * 2, hipMemcpyHostToDevice));
      MS_HIP(hipMemcpy(base + b1 + b2, s2k.data(), (size_t)n_cols8 * 2, hipMemcpyHostToDevice));
      MS_HIP(hipMemcpy(base + b1 + 2 * b2, kap.data(), b4, hipMemcpyHostToDevice));
      it = ctx->gen_tables.emplace(key, blob).first;
    }
    const int n_cols = L / 2 + 1, n_cols8 = (n_cols + kSlab - 1) / kSlab * kSlab;
    const size_t b1 = ((size_t)(n_cols8 / kSlab) * 2 + 15) / 16 * 16, b2 = ((size_t)n_cols8 * 2 + 15) / 16 * 16;
    char* base = reinterpret_cast<char*>(it->second);
    g->Sr = S + (S & 1);
    g->n_cols8 = n_cols8;
    g->ups_slabs = reinterpret_cast<const uint16_t*>(base);
    g->ups_pos = reinterpret_cast<const uint16_t*>(base + b1);
    g->ups_identity = reinterpret_cast<const uint16_t*>(base + b1 + b2);
    g->kappa = reinterpret_cast<const KappaEntry*>(base + b1 + 2 * b2);
  }
  return MISTITCH_OK;
}

// Either decomposition of one canonical strip shape, and what it needs from the workspace.
struct AnyShape {
  bool general;
  ShapeCfg fast;
  GenShape gen;
  int S, L;
  int nblk;           // blocks of the last pass = partial peak lists per pair
  int n_stored;       // columns the subpixel stage works on
  bool f64;           // precision of the Hermitian path for this shape
  size_t spec_bytes;  // per pair, in RegWorkspace::spectrum
  size_t slab_bytes;  // per pair, in RegWorkspace::slabs
};

// Power-of-two sides take the Hermitian path; so does a strip whose LONG side alone is a power of two (the
// usual case of 2048-pixel tiles with an arbitrary edge width), with chirp-z transforms on the short axis
// only and in complex128 whatever precision was asked for.
static bool fast_shape_ok(int S, int L) {
  return is_pow2(L) && L >= 8 && L <= 4096 && S <= 1024 && (is_pow2(S) ? S >= 8 : S >= 3);
}

static int make_any_shape(mistitch_ctx* ctx, int S, int L, bool f64, AnyShape* a) {
  a->S = S;
  a->L = L;
  a->general = !fast_shape_ok(S, L);
  int rc;
  if (a->general) {
    if ((rc = make_gen_shape(ctx, S, L, &a->gen))) return rc;
    a->nblk = S;
    a->n_stored = a->gen.n_cols8;
    a->spec_bytes = a->slab_bytes = (size_t)S * L * 16;
  } else {
    a->f64 = f64 || !is_pow2(S);
    if ((rc = make_shape(ctx, S, L, a->f64, &a->fast))) return rc;
    const size_t esz = a->f64 ? 16 : 8;
    a->nblk = (S + 2 * a->fast.NF_inv - 1) / (2 * a->fast.NF_inv);
    a->n_stored = a->fast.n_stored;
    a->spec_bytes = (size_t)S * (L + 128 / esz) * esz;
    a->slab_bytes = (size_t)slab_pair_elems(S, L) * esz;
  }
  return MISTITCH_OK;
}

struct RegWorkspace {
  PairDesc* desc;
  uint16_t* staging;
  void* spectrum;  // [pairs][S][L + pad] rows: output of the column pass
  void* slabs;     // [pairs][L / 8][S + 1][8]: output of the first long-axis pass
  Peak* partial;
  Peak* peaks;
  Cand* cand;
  unsigned long long* acc;
  int partial_stride;
  int n_cand = kNumCand;  // candidates per pair: 16 per peak the search consumes
  bool window = false;    // MISTITCH_REG_WINDOW_HANN
  // subpixel stage (n_up = 0: off)
  int n_up = 0;
  double uf = 0.0;
  UpsState* ups = nullptr;        // [pairs]
  Peak* partial_abs = nullptr;    // [chunk][abs_stride]
  int abs_stride = 0;
  Peak* partial_ups = nullptr;    // [chunk][n_row_blocks]
  int n_row_blocks = 0;
  double2* nq = nullptr;          // [chunk][n_stored_max]
  double2* gxy = nullptr;         // [chunk][kUpsLongParts][kUpsNodes] Nyquist-row sums per long-axis node displacement, by part of the column range
  double2* t1 = nullptr;          // [chunk][n_stored_max][kUpsNodes]: first-stage results by |frequency| and node
  double2* ccn = nullptr;         // [chunk][kUpsLongParts][kUpsNodes][kUpsNodes]: the correlation at the node displacements, by part
  const double* ups_node = nullptr;  // [kUpsNodes] node displacements relative to the coarse shift, pixels
  const double* ups_p = nullptr;     // [kUpsMax][kUpsNodes] interpolation matrix, nodes -> upsampled grid
  const double* ups_pt = nullptr;    // [kUpsNodes][kUpsMax] its transpose
  int n_stored_max = 0;
};

// Node displacements and interpolation matrix of the subpixel stage for one upsample factor (device tables, built on
// first use).  The grid offsets are o_u = (u - floor(n_up / 2)) / uf, u < n_up; on the interval they span, scaled
// to [-1, 1], the nodes are the Chebyshev points of the first kind t_j = cos(pi (2 j + 1) / (2 M)) and
// P[u][j] = (w_j / (x_u - t_j)) / sum_k (w_k / (x_u - t_k)), w_j = (-1)^j sin(pi (2 j + 1) / (2 M)), is the barycentric
// form of the interpolating polynomial's value at x_u.
static int get_ups_grid(mistitch_ctx* ctx, int uf, int n_up, RegWorkspace* ws) {
  auto it = ctx->ups_grid_tables.find(uf);
  if (it == ctx->ups_grid_tables.end()) {
    constexpr int M = kUpsNodes;
    const long double pi = 3.141592653589793238462643383279502884L;
    std::vector<double> blob((size_t)M + 2 * (size_t)kUpsMax * M, 0.0);
    double* node = blob.data();
    double* P = node + M;
    double* Pt = P + (size_t)kUpsMax * M;
    const long double centre = ((long double)(n_up - 1) / 2 - (long double)(n_up / 2)) / uf;
    const long double half = (long double)(n_up - 1) / (2 * (long double)uf);
    long double t[M], wgt[M];
    for (int j = 0; j < M; ++j) {
      t[j] = cosl(pi * (2 * j + 1) / (2 * M));
      wgt[j] = ((j & 1) ? -1 : 1) * sinl(pi * (2 * j + 1) / (2 * M));
      node[j] = (double)(centre + half * t[j]);
    }
    for (int u = 0; u < n_up; ++u) {
      const long double x = (2 * (long double)u - (n_up - 1)) / (long double)(n_up - 1);
      int hit = -1;
      long double q[M], sum = 0;
      for (int j = 0; j < M; ++j) {
        if (x == t[j]) {
          hit = j;
          break;
        }
        q[j] = wgt[j] / (x - t[j]);
        sum += q[j];
      }
      for (int j = 0; j < M; ++j) {
        const double v = hit >= 0 ? (j == hit ? 1.0 : 0.0) : (double)(q[j] / sum);
        P[(size_t)u * M + j] = v;
        Pt[(size_t)j * kUpsMax + u] = v;
      }
    }
    void* dev = nullptr;
    MS_HIP(hipMalloc(&dev, blob.size() * sizeof(double)));
    ctx->shape_allocs.push_back(dev);
    MS_HIP(hipMemcpy(dev, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->ups_grid_tables.emplace(uf, dev).first;
  }
  const double* base = reinterpret_cast<const double*>(it->second);
  ws->ups_node = base;
  ws->ups_p = base + kUpsNodes;
  ws->ups_pt = base + kUpsNodes + (size_t)kUpsMax * kUpsNodes;
  return MISTITCH_OK;
}

// Carve the subpixel buffers out of the context's workspace.
static int reserve_subpixel(mistitch_ctx* ctx, int n_pairs, int chunk, int abs_stride, int n_stored_max, int uf,
                            RegWorkspace* ws) {
  const int n_up = (3 * uf + 1) / 2;  // ceil(1.5 uf)
  MS_REQUIRE(uf >= 2 && n_up <= kUpsMax, MISTITCH_ERR_UNSUPPORTED, "upsample factor %d outside [2, %d]", uf,
             2 * kUpsMax / 3);
  const int n_row_blocks = (n_up + kUpsRows - 1) / kUpsRows;
  auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_ups = pad((size_t)n_pairs * sizeof(UpsState));
  const size_t b_abs = pad((size_t)chunk * abs_stride * sizeof(Peak));
  const size_t b_pu = pad((size_t)chunk * n_row_blocks * sizeof(Peak));
  const size_t b_nq = pad((size_t)chunk * n_stored_max * sizeof(double2));
  const size_t b_g = pad((size_t)chunk * kUpsLongParts * kUpsNodes * sizeof(double2));
  const size_t b_t1 = pad((size_t)chunk * kUpsNodes * n_stored_max * sizeof(double2));
  const size_t b_cc = pad((size_t)chunk * kUpsLongParts * kUpsNodes * kUpsNodes * sizeof(double2));
  int rc;
  if ((rc = get_ups_grid(ctx, uf, n_up, ws))) return rc;
  if ((rc = ctx->ups.reserve(b_ups + b_abs + b_pu + b_nq + b_g + b_t1 + b_cc))) return rc;
  char* base = reinterpret_cast<char*>(ctx->ups.ptr);
  ws->n_up = n_up;
  ws->uf = (double)uf;
  ws->ups = reinterpret_cast<UpsState*>(base);
  ws->partial_abs = reinterpret_cast<Peak*>(base + b_ups);
  ws->abs_stride = abs_stride;
  ws->partial_ups = reinterpret_cast<Peak*>(base + b_ups + b_abs);
  ws->n_row_blocks = n_row_blocks;
  ws->nq = reinterpret_cast<double2*>(base + b_ups + b_abs + b_pu);
  ws->gxy = reinterpret_cast<double2*>(base + b_ups + b_abs + b_pu + b_nq);
  ws->t1 = reinterpret_cast<double2*>(base + b_ups + b_abs + b_pu + b_nq + b_g);
  ws->ccn = reinterpret_cast<double2*>(base + b_ups + b_abs + b_pu + b_nq + b_g + b_t1);
  ws->n_stored_max = n_stored_max;
  return MISTITCH_OK;
}

// The strips of ALL pairs of a call are staged by one launch, and their candidate overlaps summed by
// one launch after the last pass, when the staging buffer for that stays modest; the transforms in
// between go in passes of `chunk` pairs, which bound their workspace.  (Six short launches of the
// candidate kernel spend a fifth of their time on their last few workgroups.)
constexpr int kStageAllPairs = 16384;  // grid dimensions of the staging kernels
static bool stage_all_pairs(const mistitch_ctx* ctx, int n_pairs, int chunk, int64_t slot_elems) {
  return n_pairs > chunk && n_pairs <= kStageAllPairs && (int64_t)n_pairs * slot_elems * 2 <= ctx->staging_limit;
}

static int reserve_registration(mistitch_ctx* ctx, int n_pairs, int chunk, int stage_slots, size_t staging_elems_per_pair,
                                size_t spectrum_bytes_per_pair, size_t slab_bytes_per_pair, int partial_stride,
                                RegWorkspace* ws, int n_cand = kNumCand) {
  int rc;
  if ((rc = ctx->desc.reserve((size_t)n_pairs * sizeof(PairDesc)))) return rc;
  if ((rc = ctx->staging.reserve((size_t)stage_slots * staging_elems_per_pair * 2 + 256))) return rc;
  if ((rc = ctx->spectrum.reserve((size_t)chunk * spectrum_bytes_per_pair))) return rc;
  if ((rc = ctx->slabs.reserve((size_t)chunk * slab_bytes_per_pair))) return rc;
  // (a call that stages every pair up front also keeps every pair's block lists, and merges them in one launch)
  if ((rc = ctx->partial.reserve((size_t)std::max(chunk, stage_slots) * partial_stride * sizeof(Peak)))) return rc;
  if ((rc = ctx->peaks.reserve((size_t)n_pairs * kMaxK * sizeof(Peak)))) return rc;
  if ((rc = ctx->cand.reserve((size_t)n_pairs * (n_cand + 1) * sizeof(Cand)))) return rc;
  if ((rc = ctx->acc.reserve((size_t)n_pairs * (n_cand + 1) * 5 * sizeof(unsigned long long)))) return rc;
  ws->n_cand = n_cand;
  ws->desc = (PairDesc*)ctx->desc.ptr;
  ws->staging = (uint16_t*)ctx->staging.ptr;
  ws->spectrum = ctx->spectrum.ptr;
  ws->slabs = ctx->slabs.ptr;
  ws->partial = (Peak*)ctx->partial.ptr;
  ws->peaks = (Peak*)ctx->peaks.ptr;
  ws->cand = (Cand*)ctx->cand.ptr;
  ws->acc = (unsigned long long*)ctx->acc.ptr;
  ws->partial_stride = partial_stride;
  return MISTITCH_OK;
}

// Column tables of the subpixel stage for one strip shape (all on the device).
struct UpsTables {
  const uint16_t* slabs;         // first position of every aligned group of 8 positions with a stored column
  int n_slabs;
  const uint16_t* pos2stored;    // storage position -> stored column
  const uint16_t* stored2kappa;  // stored column -> |frequency| (0xffff: unused)
  const KappaEntry* kappa;       // |frequency| -> stored column, sign, self-conjugate
};

// The subpixel stage for pairs [pair0, pair0 + n) whose normalised spectra lie in `rslabs` (slabs of Sr rows
// in ascending signed frequency, `pair_stride` elements per pair) and whose |PCM| maxima per block are in
// ws.partial_abs (`nblk` per pair).
template <typename T>
static int launch_subpixel(mistitch_ctx* ctx, const UpsTables& tb, const RegWorkspace& ws, const int8_t* status, int pair0,
                           int n, int want_transposed, int S, int L, int Sr, int64_t pair_stride, const cx<T>* rslabs,
                           int nblk, int k) {
  int rc;
  MS_REQUIRE(k >= 1 && k <= kMaxK, MISTITCH_ERR_UNSUPPORTED, "the subpixel stage needs a peak search");
  MS_LAUNCH(ctx, "k_ups_coarse", k_ups_coarse, dim3(n), dim3(kThreads), 0, ws.desc, status, pair0, S, L, want_transposed,
            ws.partial_abs, ws.abs_stride, nblk, ws.ups);
  // slabs per workgroup: two, so that two workgroups per CU overlap each other's staging (four measured
  // slower); one where LDS does not hold two
  const size_t slab_lds = (size_t)Sr * 2 * kSlab * sizeof(double);
#ifndef MS_UPS_NS
#define MS_UPS_NS 2
#endif
  int ns_ups = MS_UPS_NS;
  while (ns_ups > 1 && ns_ups * slab_lds > 144 * 1024) ns_ups >>= 1;
  MS_REQUIRE(slab_lds <= kLdsMax, MISTITCH_ERR_UNSUPPORTED, "short side %d too long for the subpixel stage", S);
#define MS_UPS_SHORT(NS)                                                                                             \
  do {                                                                                                               \
    const size_t ups_lds = std::max(NS * slab_lds, (size_t)NS * 4096); /* the slabs, then the sums of half the waves */ \
    if ((rc = allow_lds(ctx, k_ups_short<T, NS>, ups_lds))) return rc;                                               \
    MS_LAUNCH(ctx, "k_ups_short", (k_ups_short<T, NS>), dim3((tb.n_slabs + NS - 1) / NS, n), dim3(kThreads),         \
              ups_lds, ws.desc, status, pair0, S, L, want_transposed, Sr, pair_stride, tb.slabs, tb.n_slabs,   \
              tb.pos2stored, tb.stored2kappa, ws.n_stored_max, rslabs, ws.ups, ws.ups_node, ws.t1, ws.nq);           \
  } while (0)
  // Slabs per workgroup of the register-resident first stage: two workgroups fit a CU (128 registers per lane, 512
  // lanes), and a launch whose workgroups all start together has no ragged last round: the pairs of a pass share
  // the device's slots evenly (a pass of 128 pairs on 256 CUs: four workgroups per pair, 33 slabs each).
#ifdef MS_UPS_GROUP
  const int ups_group = MS_UPS_GROUP;
#else
  const int ups_wgs_per_pair = std::max(1, std::min(tb.n_slabs, 2 * ctx->n_cus / std::max(1, n)));
  const int ups_group = std::min(kUpsGroupMax, std::max(1, (tb.n_slabs + ups_wgs_per_pair - 1) / ups_wgs_per_pair));
#endif
#define MS_UPS_SHORT_REG(QS)                                                                                         \
  do {                                                                                                               \
    const size_t reg_lds = slab_lds + 8 * 4 * 64 * sizeof(double) + kUpsGroupMax * (1 + 2 * kSlab) * sizeof(uint16_t); \
    if ((rc = allow_lds(ctx, k_ups_short_reg<T, QS>, reg_lds))) return rc;                                           \
    MS_LAUNCH(ctx, "k_ups_short", (k_ups_short_reg<T, QS>), dim3((tb.n_slabs + ups_group - 1) / ups_group, n),       \
              dim3(kUpsShortThreads), reg_lds, ws.desc, status, pair0, S, L, want_transposed, Sr, pair_stride,       \
              tb.slabs, tb.n_slabs, tb.pos2stored, tb.stored2kappa, ws.n_stored_max, rslabs, ws.ups, ws.ups_node,    \
              ws.t1, ws.nq, ups_group);                                                                              \
  } while (0)
  if (tb.n_slabs > 0) {
    // short sides up to 512 keep E in registers (32 or 64 steps per quarter of the kk range)
    if (Sr <= 256) MS_UPS_SHORT_REG(32);
    else if (Sr <= 512) MS_UPS_SHORT_REG(64);
    else if (ns_ups == 2) MS_UPS_SHORT(2);
    else MS_UPS_SHORT(1);
  }
#undef MS_UPS_SHORT
#undef MS_UPS_SHORT_REG
  MS_LAUNCH(ctx, "k_ups_long", k_ups_long, dim3(kUpsLongParts, n), dim3(kUpsLongThreads), 0, ws.desc, status, pair0, S, L,
            want_transposed, tb.kappa, ws.n_stored_max, ws.t1, ws.nq, ws.ups, ws.ups_node, ws.ccn, ws.gxy);
  MS_LAUNCH(ctx, "k_ups_interp", k_ups_interp, dim3(ws.n_row_blocks, n), dim3(kThreads), 0, ws.desc, status, pair0, S, L,
            want_transposed, ws.ccn, ws.gxy, ws.ups_node, ws.ups_p, ws.ups_pt, ws.ups, ws.n_up, ws.partial_ups,
            ws.n_row_blocks);
  return MISTITCH_OK;
}

// Device table of the symmetric Hann window of length n, w[i] = 0.5 - 0.5 cos(2 pi i / (n - 1)) (numpy.hanning;
// 1 for n = 1), built on first use.
static int get_hann(mistitch_ctx* ctx, int n, const double** out) {
  auto it = ctx->hann_tables.find(n);
  if (it == ctx->hann_tables.end()) {
    std::vector<double> w((size_t)n);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < n; ++i) w[i] = n == 1 ? 1.0 : 0.5 + 0.5 * cos(pi * (double)(1 - n + 2 * i) / (double)(n - 1));
    void* dev = nullptr;
    MS_HIP(hipMalloc(&dev, (size_t)n * sizeof(double)));
    ctx->shape_allocs.push_back(dev);
    MS_HIP(hipMemcpy(dev, w.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->hann_tables.emplace(n, dev).first;
  }
  *out = reinterpret_cast<const double*>(it->second);
  return MISTITCH_OK;
}

// Forward long-axis pass on the register-resident engine, for the line lengths it is built for (complex128 plans
// whose passes are (L / 512, 8, 8, 8)); *done = false where the caller has to take the in-LDS kernel.
template <int N>
static int launch_long_fwd_reg_n(mistitch_ctx* ctx, const ShapeCfg& cfg, const RegWorkspace& ws, const int8_t* status,
                                 int pair0, int n, int want_transposed, double2* slabs) {
  const size_t lds = (size_t)RegFft<N>::kSlots * sizeof(double2);
  const double2* roots = reinterpret_cast<const double2*>(cfg.planL.tw);
  int rc;
  if (ws.window) {
    const double *win_s = nullptr, *win_l = nullptr;
    if ((rc = get_hann(ctx, cfg.S, &win_s))) return rc;
    if ((rc = get_hann(ctx, cfg.L, &win_l))) return rc;
    if ((rc = allow_lds(ctx, k_fft_long_fwd_reg<N, true>, lds))) return rc;
    MS_LAUNCH(ctx, "k_fft_long_fwd", (k_fft_long_fwd_reg<N, true>), dim3(cfg.S, n), dim3(N / 8), lds, ws.desc, status, pair0,
              cfg.S, want_transposed, roots, slabs, win_s, win_l);
  } else {
    if ((rc = allow_lds(ctx, k_fft_long_fwd_reg<N, false>, lds))) return rc;
    MS_LAUNCH(ctx, "k_fft_long_fwd", (k_fft_long_fwd_reg<N, false>), dim3(cfg.S, n), dim3(N / 8), lds, ws.desc, status, pair0,
              cfg.S, want_transposed, roots, slabs, (const double*)nullptr, (const double*)nullptr);
  }
  return MISTITCH_OK;
}
static int launch_long_fwd_reg(mistitch_ctx* ctx, const ShapeCfg& cfg, const RegWorkspace& ws, const int8_t* status,
                               int pair0, int n, int want_transposed, double2* slabs, bool* done) {
  *done = true;
  if (kLongMaxRadix<double> == 8 && cfg.S <= 65535) {
    switch (cfg.L) {
      case 512: return launch_long_fwd_reg_n<512>(ctx, cfg, ws, status, pair0, n, want_transposed, slabs);
      case 1024: return launch_long_fwd_reg_n<1024>(ctx, cfg, ws, status, pair0, n, want_transposed, slabs);
      case 2048: return launch_long_fwd_reg_n<2048>(ctx, cfg, ws, status, pair0, n, want_transposed, slabs);
      case 4096: return launch_long_fwd_reg_n<4096>(ctx, cfg, ws, status, pair0, n, want_transposed, slabs);
      default: break;
    }
  }
  *done = false;
  return MISTITCH_OK;
}

// The three FFT kernels for pairs [pair0, pair0 + n) of one canonical shape.
template <typename T>
static int launch_fft(mistitch_ctx* ctx, const ShapeCfg& cfg, const RegWorkspace& ws, const int8_t* status,
                      int pair0, int n, int want_transposed, bool vec_ok, int k, double* pcm_out) {
  cx<T>* spec = reinterpret_cast<cx<T>*>(ws.spectrum);
  cx<T>* slabs = reinterpret_cast<cx<T>*>(ws.slabs);
  int rc;
  bool fwd_done = false;
  if constexpr (std::is_same<T, double>::value) {
    if (!getenv("MISTITCH_FFT_LDS_ENGINE") &&
        (rc = launch_long_fwd_reg(ctx, cfg, ws, status, pair0, n, want_transposed, slabs, &fwd_done)))
      return rc;
  }
  if (fwd_done) {
  } else if (ws.window) {
    const double *win_s = nullptr, *win_l = nullptr;
    if ((rc = get_hann(ctx, cfg.S, &win_s))) return rc;
    if ((rc = get_hann(ctx, cfg.L, &win_l))) return rc;
    if (vec_ok) {
      if ((rc = allow_lds(ctx, k_fft_long_fwd<T, true, true>, cfg.lds_fwd))) return rc;
      MS_LAUNCH(ctx, "k_fft_long_fwd", (k_fft_long_fwd<T, true, true>), dim3(cfg.S / cfg.NF_fwd, n), dim3(kThreads),
                cfg.lds_fwd, ws.desc, status, pair0, cfg.S, cfg.L, want_transposed, cfg.planL, cfg.NF_fwd, slabs, win_s, win_l);
    } else {
      if ((rc = allow_lds(ctx, k_fft_long_fwd<T, false, true>, cfg.lds_fwd))) return rc;
      MS_LAUNCH(ctx, "k_fft_long_fwd", (k_fft_long_fwd<T, false, true>), dim3(cfg.S / cfg.NF_fwd, n), dim3(kThreads),
                cfg.lds_fwd, ws.desc, status, pair0, cfg.S, cfg.L, want_transposed, cfg.planL, cfg.NF_fwd, slabs, win_s, win_l);
    }
  } else if (vec_ok) {
    if ((rc = allow_lds(ctx, k_fft_long_fwd<T, true>, cfg.lds_fwd))) return rc;
    MS_LAUNCH(ctx, "k_fft_long_fwd", (k_fft_long_fwd<T, true>), dim3(cfg.S / cfg.NF_fwd, n), dim3(kThreads), cfg.lds_fwd,
                       ws.desc, status, pair0, cfg.S, cfg.L, want_transposed, cfg.planL, cfg.NF_fwd, slabs,
                       (const double*)nullptr, (const double*)nullptr);
  } else {
    if ((rc = allow_lds(ctx, k_fft_long_fwd<T, false>, cfg.lds_fwd))) return rc;
    MS_LAUNCH(ctx, "k_fft_long_fwd", (k_fft_long_fwd<T, false>), dim3(cfg.S / cfg.NF_fwd, n), dim3(kThreads), cfg.lds_fwd,
                       ws.desc, status, pair0, cfg.S, cfg.L, want_transposed, cfg.planL, cfg.NF_fwd, slabs,
                       (const double*)nullptr, (const double*)nullptr);
  }
  const T eps = (T)(std::numeric_limits<double>::epsilon() * 100.0);
  // integer strips: the self-conjugate frequencies are rounded to their exact values (MISTITCH_NO_EXACT_BINS: a diagnostic switch)
  const int exact_bins = (ws.window || getenv("MISTITCH_NO_EXACT_BINS")) ? 0 : 1;
  bool column_pass_done = false;
  if constexpr (std::is_same<T, double>::value) {
    if (cfg.mixed_short) {
      constexpr int kNT = kMixedShortThreads;
      if ((rc = allow_lds(ctx, k_fft_short_cps<double, kNT, 8, true>, cfg.lds_cps))) return rc;
      MS_LAUNCH(ctx, "k_fft_short_cps", (k_fft_short_cps<double, kNT, 8, true>), dim3(cfg.n_units, n), dim3(kNT), cfg.lds_cps,
                ws.desc, status, pair0, cfg.S, cfg.L, want_transposed, cfg.planS, cfg.planL, cfg.units, 2 * cfg.C, slabs, spec,
                eps, ws.n_up ? slabs : (cx<T>*)nullptr, cfg.pos2stored, cfg.n_stored, exact_bins);
      column_pass_done = true;
    } else if (cfg.gen_short) {
      if ((rc = allow_lds(ctx, k_fft_short_cps_gen, cfg.lds_cps))) return rc;
      MS_LAUNCH(ctx, "k_fft_short_cps", k_fft_short_cps_gen, dim3(cfg.n_units, n), dim3(kGenShortThreads), cfg.lds_cps, ws.desc, status,
                pair0, cfg.S, cfg.L, want_transposed, cfg.gpS, cfg.planL, cfg.units, 2 * cfg.C, slabs, spec, eps,
                ws.n_up ? slabs : (cx<T>*)nullptr, cfg.pos2stored, cfg.n_stored, exact_bins);
      column_pass_done = true;
    }
  }
  if constexpr (std::is_same<T, double>::value) {
    if (!column_pass_done && !cfg.gen_short && cfg.reg_cols && (cfg.S == 128 || cfg.S == 256) && kShortMaxRadix<double> == 8 &&
        !getenv("MISTITCH_FFT_LDS_ENGINE")) {
      const double2* roots = reinterpret_cast<const double2*>(cfg.planS.tw);
#define MS_CPS_REG(SS)                                                                                                 \
  do {                                                                                                                 \
    const size_t lds8 = (size_t)2 * (SS + SS / 8) * 8 * sizeof(double2), lds4 = lds8 / 2;                               \
    double2* keep = ws.n_up ? slabs : (double2*)nullptr;                                                               \
    if ((rc = allow_lds(ctx, k_fft_short_cps_reg<SS, 8>, lds8))) return rc;                                            \
    if (!getenv("MISTITCH_CPS_HALF_UNITS")) {  /* whole units by default: halves measured 2 % slower */                                                                          \
      MS_LAUNCH(ctx, "k_fft_short_cps", (k_fft_short_cps_reg<SS, 8>), dim3(cfg.n_units, n), dim3(2 * SS), lds8, ws.desc, \
                status, pair0, cfg.L, want_transposed, roots, cfg.planS.mpos, cfg.planS.pos2freq, cfg.planL.mpos,       \
                cfg.units, cfg.n_units, slabs, spec, (double)eps, keep, cfg.pos2stored, cfg.n_stored, exact_bins);                  \
    } else {                                                                                                           \
      if ((rc = allow_lds(ctx, k_fft_short_cps_reg<SS, 4>, lds4))) return rc;                                          \
      if (cfg.n_paired > 0)                                                                                            \
        MS_LAUNCH(ctx, "k_fft_short_cps", (k_fft_short_cps_reg<SS, 4>), dim3(16 * ((cfg.n_paired + 7) / 8), n), dim3(SS), \
                  lds4, ws.desc, status, pair0, cfg.L, want_transposed, roots, cfg.planS.mpos, cfg.planS.pos2freq,     \
                  cfg.planL.mpos, cfg.units_paired, cfg.n_paired, slabs, spec, (double)eps, keep, cfg.pos2stored,      \
                  cfg.n_stored, exact_bins);                                                                           \
      if (cfg.n_single > 0)                                                                                            \
        MS_LAUNCH(ctx, "k_fft_short_cps", (k_fft_short_cps_reg<SS, 8>), dim3(cfg.n_single, n), dim3(2 * SS), lds8,     \
                  ws.desc, status, pair0, cfg.L, want_transposed, roots, cfg.planS.mpos, cfg.planS.pos2freq,           \
                  cfg.planL.mpos, cfg.units_single, cfg.n_single, slabs, spec, (double)eps, keep, cfg.pos2stored,      \
                  cfg.n_stored, exact_bins);                                                                           \
    }                                                                                                                  \
  } while (0)
      if (cfg.S == 128) MS_CPS_REG(128);
      else MS_CPS_REG(256);
#undef MS_CPS_REG
      column_pass_done = true;
    }
  }
  if (!column_pass_done) {
    MS_REQUIRE(!cfg.gen_short, MISTITCH_ERR_UNSUPPORTED, "a short side of %d needs the complex128 path", cfg.S);
    constexpr int kNT = kShortThreads<T>, kSR = kShortMaxRadix<T>;
    if ((rc = allow_lds(ctx, k_fft_short_cps<T, kNT, kSR>, cfg.lds_cps))) return rc;
    MS_LAUNCH(ctx, "k_fft_short_cps", (k_fft_short_cps<T, kNT, kSR>), dim3(cfg.n_units, n), dim3(kNT), cfg.lds_cps, ws.desc,
              status, pair0, cfg.S, cfg.L, want_transposed, cfg.planS, cfg.planL, cfg.units, 2 * cfg.C, slabs, spec, eps,
              ws.n_up ? slabs : (cx<T>*)nullptr, cfg.pos2stored, cfg.n_stored, exact_bins);
  }
  // Where the last radix pass ranks its values (two peaks, no PCM plane), a workgroup takes several blocks of lines
  // in turn and reduces once - as long as the launch still fills the chip a few times over.
  const int nblk_inv = (cfg.S + 2 * cfg.NF_inv - 1) / (2 * cfg.NF_inv);
  int seq = 1;
  if (k > 0 && k <= 2 && pcm_out == nullptr)   // (more peaks: one block per workgroup)
    while (seq < kLongInvSeq && (int64_t)((nblk_inv + 2 * seq - 1) / (2 * seq)) * n >= 2048) seq *= 2;
  bool inv_done = false;
  if constexpr (std::is_same<T, double>::value) {
    // two peaks at most and no correlation plane: the register-resident engine, for the line lengths it is built for
    if (k > 0 && k <= kMaxK && pcm_out == nullptr && cfg.NF_inv == 1 && kLongMaxRadix<double> == 8 && !getenv("MISTITCH_FFT_LDS_ENGINE")) {
      const bool wide_k = k > 2;
      const double2* roots = reinterpret_cast<const double2*>(cfg.planL.tw);
      Peak* pabs = ws.n_up ? ws.partial_abs : (Peak*)nullptr;
#define MS_INV_REG(NN)                                                                                                \
  do {                                                                                                                \
    const size_t lds = (size_t)RegFft<NN>::kSlots * sizeof(double2);                                                  \
    if (wide_k && pabs != nullptr) {                                                                                  \
      if ((rc = allow_lds(ctx, k_fft_long_inv_reg<NN, true, true>, lds))) return rc;                                  \
      MS_LAUNCH(ctx, "k_fft_long_inv", (k_fft_long_inv_reg<NN, true, true>), dim3(nblk_inv, n), dim3(NN / 8), lds,     \
                ws.desc, status, pair0, cfg.S, want_transposed, roots, cfg.stored_pos, cfg.mirror_fill, cfg.n_stored, \
                spec, k, ws.partial, ws.partial_stride, pabs, ws.abs_stride, nblk_inv, 1);                            \
    } else if (wide_k) {                                                                                              \
      if ((rc = allow_lds(ctx, k_fft_long_inv_reg<NN, false, true>, lds))) return rc;                                 \
      MS_LAUNCH(ctx, "k_fft_long_inv", (k_fft_long_inv_reg<NN, false, true>), dim3(nblk_inv, n), dim3(NN / 8), lds,    \
                ws.desc, status, pair0, cfg.S, want_transposed, roots, cfg.stored_pos, cfg.mirror_fill, cfg.n_stored, \
                spec, k, ws.partial, ws.partial_stride, pabs, ws.abs_stride, nblk_inv, 1);                            \
    } else if (pabs != nullptr) {                                                                                     \
      if ((rc = allow_lds(ctx, k_fft_long_inv_reg<NN, true>, lds))) return rc;                                        \
      MS_LAUNCH(ctx, "k_fft_long_inv", (k_fft_long_inv_reg<NN, true>), dim3((nblk_inv + seq - 1) / seq, n), dim3(NN / 8), lds, \
                ws.desc, status, pair0, cfg.S, want_transposed, roots, cfg.stored_pos, cfg.mirror_fill, cfg.n_stored, \
                spec, k, ws.partial, ws.partial_stride, pabs, ws.abs_stride, nblk_inv, seq);                          \
    } else {                                                                                                          \
      if ((rc = allow_lds(ctx, k_fft_long_inv_reg<NN, false>, lds))) return rc;                                       \
      MS_LAUNCH(ctx, "k_fft_long_inv", (k_fft_long_inv_reg<NN, false>), dim3((nblk_inv + seq - 1) / seq, n), dim3(NN / 8), lds, \
                ws.desc, status, pair0, cfg.S, want_transposed, roots, cfg.stored_pos, cfg.mirror_fill, cfg.n_stored, \
                spec, k, ws.partial, ws.partial_stride, pabs, ws.abs_stride, nblk_inv, seq);                          \
    }                                                                                                                 \
    inv_done = true;                                                                                                  \
  } while (0)
      if (cfg.L == 512) MS_INV_REG(512);
      else if (cfg.L == 1024) MS_INV_REG(1024);
      else if (cfg.L == 2048) MS_INV_REG(2048);
      else if (cfg.L == 4096) MS_INV_REG(4096);
#undef MS_INV_REG
    }
  }
  if (!inv_done) {
    if ((rc = allow_lds(ctx, k_fft_long_inv<T>, cfg.lds_inv))) return rc;
    MS_LAUNCH(ctx, "k_fft_long_inv", (k_fft_long_inv<T>), dim3((nblk_inv + seq - 1) / seq, n), dim3(kThreads), cfg.lds_inv, ws.desc,
              status, pair0, cfg.S, cfg.L, want_transposed, cfg.planL, cfg.stored_pos, cfg.mirror_fill, cfg.n_stored, cfg.NF_inv,
              spec, k, ws.partial, ws.partial_stride, pcm_out, ws.n_up ? ws.partial_abs : (Peak*)nullptr, ws.abs_stride,
              nblk_inv, seq);
  }
  if (ws.n_up) {
    // subpixel refinement of these pairs, while their normalised spectra still sit in the slab buffer
    UpsTables tabs{cfg.ups_slabs, cfg.n_ups_slabs, cfg.pos2stored, cfg.stored2kappa, cfg.kappa};
    if ((rc = launch_subpixel<T>(ctx, tabs, ws, status, pair0, n, want_transposed, cfg.S, cfg.L, cfg.S + (cfg.S & 1),
                                 slab_pair_elems(cfg.S, cfg.L), slabs, (cfg.S + 2 * cfg.NF_inv - 1) / (2 * cfg.NF_inv), k)))
      return rc;
  }
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

// The chirp-z pipeline for pairs [pair0, pair0 + n) of one canonical shape.
static int launch_fft_general(mistitch_ctx* ctx, const GenShape& g, const RegWorkspace& ws, const int8_t* status,
                              int pair0, int n, int want_transposed, int k, double* pcm_out) {
  double2* Zf = reinterpret_cast<double2*>(ws.slabs);
  double2* R = reinterpret_cast<double2*>(ws.spectrum);
  int rc;
  if ((rc = allow_lds(ctx, k_gen_rows_fwd, g.lds_rows))) return rc;
  if ((rc = allow_lds(ctx, k_gen_cols, g.lds_cols))) return rc;
  if ((rc = allow_lds(ctx, k_gen_rows_inv, g.lds_rows))) return rc;
  const double eps = std::numeric_limits<double>::epsilon() * 100.0;
  const int col_blocks = (g.L + g.C - 1) / g.C;
  const double *win_s = nullptr, *win_l = nullptr;
  if (ws.window) {
    if ((rc = get_hann(ctx, g.S, &win_s))) return rc;
    if ((rc = get_hann(ctx, g.L, &win_l))) return rc;
  }
  MS_LAUNCH(ctx, "k_gen_rows_fwd", k_gen_rows_fwd, dim3(g.S, n), dim3(kThreads), g.lds_rows, ws.desc, status, pair0, g.S, g.L,
            want_transposed, g.pl, Zf, win_s, win_l);
  MS_LAUNCH(ctx, "k_gen_cols", k_gen_cols, dim3(col_blocks, n), dim3(kThreads), g.lds_cols, ws.desc, status, pair0, g.S, g.L,
            want_transposed, g.ps, g.C, 0, Zf);
  const int cps_blocks = std::max(1, std::min(1024, (g.S * g.L + kThreads - 1) / kThreads));
  MS_LAUNCH(ctx, "k_gen_cps", k_gen_cps, dim3(cps_blocks, n), dim3(kThreads), 0, ws.desc, status, pair0, g.S, g.L,
            want_transposed, Zf, R, eps, ws.window ? 0 : 1);
  const int64_t keep_stride = slab_pair_elems(g.Sr, g.n_cols8);
  if (ws.n_up) {
    // the forward spectra are no longer needed: their buffer takes the copy of R the subpixel stage reads
    MS_REQUIRE((size_t)keep_stride <= (size_t)g.S * g.L, MISTITCH_ERR_UNSUPPORTED,
               "strip %d x %d too small for the subpixel stage", g.S, g.L);
    const int keep_blocks = std::max(1, std::min(256, (g.Sr * g.n_cols8 + kThreads - 1) / kThreads));
    MS_LAUNCH(ctx, "k_gen_keep_r", k_gen_keep_r, dim3(keep_blocks, n), dim3(kThreads), 0, ws.desc, status, pair0, g.S, g.L,
              want_transposed, R, g.Sr, g.n_cols8, keep_stride, Zf);
  }
  MS_LAUNCH(ctx, "k_gen_cols", k_gen_cols, dim3(col_blocks, n), dim3(kThreads), g.lds_cols, ws.desc, status, pair0, g.S, g.L,
            want_transposed, g.ps, g.C, 1, R);
  MS_LAUNCH(ctx, "k_gen_rows_inv", k_gen_rows_inv, dim3(g.S, n), dim3(kThreads), g.lds_rows, ws.desc, status, pair0, g.S, g.L,
            want_transposed, g.pl, R, k, ws.partial, ws.partial_stride, pcm_out,
            ws.n_up ? ws.partial_abs : (Peak*)nullptr, ws.abs_stride);
  if (ws.n_up) {
    UpsTables tabs{g.ups_slabs, g.n_cols8 / kSlab, g.ups_pos, g.ups_identity, g.kappa};
    if ((rc = launch_subpixel<double>(ctx, tabs, ws, status, pair0, n, want_transposed, g.S, g.L, g.Sr, keep_stride, Zf,
                                      g.S, k)))
      return rc;
  }
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

static int launch_any(mistitch_ctx* ctx, const AnyShape& a, bool f64, const RegWorkspace& ws, const int8_t* status,
                      int pair0, int n, int want_transposed, bool vec_ok, int k, double* pcm_out) {
  if (a.general) {
    return launch_fft_general(ctx, a.gen, ws, status, pair0, n, want_transposed, k, pcm_out);
  }
  return a.f64 ? launch_fft<double>(ctx, a.fast, ws, status, pair0, n, want_transposed, vec_ok, k, pcm_out)
               : launch_fft<float>(ctx, a.fast, ws, status, pair0, n, want_transposed, vec_ok, k, pcm_out);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mistitch

using namespace mistitch;

#ifdef MS_STAMPS
extern "C" int mistitch_debug_stamps(unsigned long long* host_out /*[3][4096][8]*/) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 3 * 4096 * 8) == hipSuccess ? 0 : -3;
}
#endif

static int register_pairs_impl(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                               int64_t tile_pitch, const void* ff, int ff_type, const int32_t* pair_idx,
                               const uint8_t* pair_dir, int n_pairs, int edge_width, unsigned flags, double* out,
                               int8_t* status, double* table = nullptr);

extern "C" int mistitch_reserve_registration(mistitch_ctx* ctx, int tile_h, int tile_w, int edge_width,
                                             int max_pairs, unsigned flags) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(tile_h > 0 && tile_w > 0 && edge_width > 0 && max_pairs > 0, MISTITCH_ERR_ARG, "non-positive size");
  MS_HIP(hipSetDevice(ctx->device));
  const bool f64 = !(flags & MISTITCH_FFT_F32);
  const int w_left = edge_width < tile_w ? edge_width : tile_w;
  const int w_top = edge_width < tile_h ? edge_width : tile_h;
  AnyShape a, b;
  int rc;
  if ((rc = make_any_shape(ctx, w_left, tile_h, f64, &a))) return rc;
  if ((rc = make_any_shape(ctx, w_top, tile_w, f64, &b))) return rc;
  const size_t spec = std::max(a.spec_bytes, b.spec_bytes);
  const int pstride = std::max(a.nblk, b.nblk) * kMaxK;
  const int chunk = max_pairs < ctx->pair_chunk ? max_pairs : ctx->pair_chunk;
  RegWorkspace ws;
  const size_t slab = std::max(a.slab_bytes, b.slab_bytes);
  // slots large enough for either direction, as a call with a flatfield stages both
  const int64_t slot_elems = 2 * std::max((int64_t)w_left * tile_h, (int64_t)w_top * tile_w);
  const int stage_slots = stage_all_pairs(ctx, max_pairs, chunk, slot_elems) ? max_pairs : chunk;
  if ((rc = reserve_registration(ctx, max_pairs, chunk, stage_slots, (size_t)slot_elems, spec, slab, pstride, &ws,
                                 (flags & (MISTITCH_REG_PEAKS10 | MISTITCH_REG_CONSENSUS)) ? kNumCandWide : kNumCand)))
    return rc;
  // Whatever else a first call sets up (tables of the subpixel stage, kernels that need their LDS limit
  // raised) is set up by making that call: max_pairs pairs of two blank tiles, left and top alternating.
  // Blank tiles are featureless, so the kernels leave early, but every launch of a real call is made.
  const size_t tile_elems = (size_t)tile_h * tile_w;
  DevBuf scratch;
  const size_t b_tiles = (2 * tile_elems * sizeof(uint16_t) + 255) / 256 * 256;
  const size_t b_idx = ((size_t)max_pairs * 2 * sizeof(int32_t) + 255) / 256 * 256;
  const size_t b_dir = ((size_t)max_pairs + 255) / 256 * 256;
  const size_t b_out = ((size_t)max_pairs * 3 * sizeof(double) + 255) / 256 * 256;
  if ((rc = scratch.reserve(b_tiles + b_idx + b_dir + b_out + (size_t)max_pairs + 256))) return rc;
  char* base = reinterpret_cast<char*>(scratch.ptr);
  std::vector<int32_t> h_idx((size_t)max_pairs * 2);
  std::vector<uint8_t> h_dir((size_t)max_pairs);
  for (int i = 0; i < max_pairs; ++i) {
    h_idx[2 * i] = 0;
    h_idx[2 * i + 1] = 1;
    h_dir[i] = (uint8_t)(i & 1);
  }
  rc = MISTITCH_OK;
  if (hipMemset(base, 0, b_tiles) != hipSuccess ||
      hipMemcpy(base + b_tiles, h_idx.data(), h_idx.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(base + b_tiles + b_idx, h_dir.data(), h_dir.size(), hipMemcpyHostToDevice) != hipSuccess)
    rc = set_error(MISTITCH_ERR_HIP, "could not set up the sizing call");
  if (rc == MISTITCH_OK)
    rc = register_pairs_impl(ctx, reinterpret_cast<const uint16_t*>(base), 2, tile_h, tile_w, (int64_t)tile_elems, nullptr,
                             MISTITCH_FF_NONE, reinterpret_cast<const int32_t*>(base + b_tiles),
                             reinterpret_cast<const uint8_t*>(base + b_tiles + b_idx), max_pairs, edge_width, flags,
                             reinterpret_cast<double*>(base + b_tiles + b_idx + b_dir),
                             reinterpret_cast<int8_t*>(base + b_tiles + b_idx + b_dir + b_out));
  if (rc == MISTITCH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
    rc = set_error(MISTITCH_ERR_HIP, "the sizing call failed");
  scratch.release();
  return rc;
}

// Body of mistitch_register_pairs / mistitch_register_pairs_flatfield (ff = NULL: tiles are used as they are).
static int register_pairs_impl(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                               int64_t tile_pitch, const void* ff, int ff_type, const int32_t* pair_idx,
                               const uint8_t* pair_dir, int n_pairs, int edge_width, unsigned flags, double* out,
                               int8_t* status, double* table) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(n_pairs >= 0 && n_tiles >= 0, MISTITCH_ERR_ARG, "negative count");
  if (n_pairs == 0) return MISTITCH_OK;
  MS_REQUIRE(tiles && pair_idx && pair_dir && out && status, MISTITCH_ERR_ARG, "NULL array argument");
  MS_REQUIRE(tile_h > 0 && tile_w > 0 && edge_width > 0, MISTITCH_ERR_ARG, "non-positive size");
  MS_REQUIRE(tile_pitch >= (int64_t)tile_h * tile_w, MISTITCH_ERR_ARG, "tile_pitch smaller than a tile");
  MS_HIP(hipSetDevice(ctx->device));
  const bool f64 = !(flags & MISTITCH_FFT_F32);
  const int H = tile_h, W = tile_w;
  const int w_left = edge_width < W ? edge_width : W;
  const int w_top = edge_width < H ? edge_width : H;
  AnyShape cfg_left, cfg_top;
  int rc;
  if ((rc = make_any_shape(ctx, w_left, H, f64, &cfg_left))) return rc;
  if ((rc = make_any_shape(ctx, w_top, W, f64, &cfg_top))) return rc;
  const bool same_shape = cfg_left.S == cfg_top.S && cfg_left.L == cfg_top.L;
  const size_t spec = std::max(cfg_left.spec_bytes, cfg_top.spec_bytes);
  const int pstride = std::max(cfg_left.nblk, cfg_top.nblk) * kMaxK;
  const int chunk = n_pairs < ctx->pair_chunk ? n_pairs : ctx->pair_chunk;
  RegWorkspace ws;
  const size_t slab = std::max(cfg_left.slab_bytes, cfg_top.slab_bytes);
  // staging slot of a pair: its two strips; with a flatfield the top strips are staged as well
  const int64_t slot_elems =
      2 * (ff != nullptr ? std::max((int64_t)w_left * H, (int64_t)w_top * W) : (int64_t)w_left * H);
  const bool stage_all = stage_all_pairs(ctx, n_pairs, chunk, slot_elems);
  const int stage_slots = stage_all ? n_pairs : chunk;
  // peaks the candidate search consumes: 2 (the reference's discrete search), or 10 (the candidate set of its
  // consensus stage, generate_translation_candidates at _translation_computation.py:876-994)
  const bool consensus = (flags & MISTITCH_REG_CONSENSUS) != 0;  // the consensus stage works on the ten-peak set
  const bool wide = (flags & MISTITCH_REG_PEAKS10) != 0 || consensus;
  const int k = wide ? 10 : 2;
  static_assert(kNumCand == 16 * 2 && kNumCandWide == 16 * 10, "16 signed interpretations per peak");
  if ((rc = reserve_registration(ctx, n_pairs, chunk, stage_slots, (size_t)slot_elems, spec, slab, pstride, &ws, 16 * k)))
    return rc;
  ws.window = (flags & MISTITCH_REG_WINDOW_HANN) != 0;
  const bool subpixel = (flags & MISTITCH_REG_SUBPIXEL) != 0;
  if (subpixel) {
    if ((rc = reserve_subpixel(ctx, n_pairs, chunk, std::max(cfg_left.nblk, cfg_top.nblk),
                               std::max(cfg_left.n_stored, cfg_top.n_stored), ctx->upsample_factor, &ws)))
      return rc;
  }
  // 16-byte loads of tile rows need aligned rows; staging lines always are.
  const bool vec_top = aligned16(tiles) && (W % 8 == 0) && (tile_pitch % 8 == 0);
  // may the top strips be read with 16-byte loads (with a flatfield they are staged too)
  const bool vec_strips = ff != nullptr ? aligned16(ws.staging) && (W % 8 == 0) && (slot_elems % 8 == 0) &&
                                              (((int64_t)w_top * W) % 8 == 0)
                                        : vec_top;
  MS_LAUNCH(ctx, "k_setup_pairs", k_setup_pairs, dim3((n_pairs + 255) / 256), dim3(256), 0, tiles, tile_pitch, H, W, w_left,
                     w_top, pair_idx, pair_dir, n_pairs, n_tiles, ws.staging, stage_slots, ws.desc, status, slot_elems,
                     ff != nullptr ? 1 : 0);
  if (ff == nullptr)
    MS_LAUNCH(ctx, "k_edge_stats", k_edge_stats, dim3(2, n_pairs), dim3(kThreads), 0, tiles, tile_pitch, H, W, w_left, w_top,
              pair_idx, pair_dir, status, 2e3);
  // strips of pairs [pair0, pair0 + n) into the staging buffer
  auto stage = [&](int pair0, int n) -> int {
    if (ff != nullptr) {
      // corrected strips of both directions go to the staging buffer; the featureless test reads them there
      const bool tiled = vec_top && H % 64 == 0 && W % 64 == 0;
      const bool fast = tiled && w_left % 64 == 0 && w_top % 64 == 0;
      if (tiled) {
        const dim3 grid(std::max(H, W) / 64, (std::max(w_left, w_top) + 63) / 64, 4 * ((n + kStagePairs - 1) / kStagePairs));
#define MS_STAGE(FF, EDGE)                                                                                           \
  MS_LAUNCH(ctx, "k_stage_strips", (k_stage_strips64<FF, EDGE>), grid, dim3(256), 0, tiles, tile_pitch, H, W, w_left, \
            w_top, pair_idx, ws.desc, status, pair0, n, ff, (unsigned)ctx->pixel_max)
        if (ff_type == MISTITCH_FF_F32) {
          if (fast) MS_STAGE(MISTITCH_FF_F32, false);
          else MS_STAGE(MISTITCH_FF_F32, true);
        } else {
          if (fast) MS_STAGE(MISTITCH_FF_F64, false);
          else MS_STAGE(MISTITCH_FF_F64, true);
        }
#undef MS_STAGE
      } else {
        const dim3 grid(256, 2 * n);
        if (ff_type == MISTITCH_FF_F32)
          MS_LAUNCH(ctx, "k_stage_strips", k_stage_strips_any<MISTITCH_FF_F32>, grid, dim3(256), 0, tiles, tile_pitch, H, W,
                    w_left, w_top, pair_idx, ws.desc, status, pair0, ff, (unsigned)ctx->pixel_max);
        else
          MS_LAUNCH(ctx, "k_stage_strips", k_stage_strips_any<MISTITCH_FF_F64>, grid, dim3(256), 0, tiles, tile_pitch, H, W,
                    w_left, w_top, pair_idx, ws.desc, status, pair0, ff, (unsigned)ctx->pixel_max);
      }
      MS_LAUNCH(ctx, "k_edge_stats", k_edge_stats_staged, dim3(2, n), dim3(kThreads), 0, ws.desc, status, pair0, 2e3);
    } else if (vec_top && H % 64 == 0 && w_left % 64 == 0) {
      MS_LAUNCH(ctx, "k_transpose_strips", k_transpose_strips64<false>, dim3(H / 64, w_left / 64, 2 * n), dim3(256), 0, tiles,
                tile_pitch, H, W, w_left, pair_idx, ws.desc, status, pair0);
    } else if (vec_top && H % 64 == 0) {
      MS_LAUNCH(ctx, "k_transpose_strips", k_transpose_strips64<true>, dim3(H / 64, (w_left + 63) / 64, 2 * n), dim3(256), 0, tiles,
                tile_pitch, H, W, w_left, pair_idx, ws.desc, status, pair0);
    } else {
      MS_LAUNCH(ctx, "k_transpose_strips", k_transpose_strips, dim3((H + 31) / 32, (w_left + 31) / 32, 2 * n), dim3(32, 8), 0,
                tiles, tile_pitch, H, W, w_left, pair_idx, ws.desc, status, pair0);
    }
    return MISTITCH_OK;
  };
  // The signed interpretations of the peaks are summed by circular sweeps when every canonical line starts on a 16-byte
  // boundary and holds whole 8-pixel groups; the subpixel slot, and everything otherwise, by windows.
#ifndef MS_NCC_SWEEP
#define MS_NCC_SWEEP 1
#endif
  const bool sweep = MS_NCC_SWEEP && H % 8 == 0 && W % 8 == 0 && aligned16(ws.staging) && slot_elems % 8 == 0 &&
                     (ff != nullptr || vec_top);
  auto candidates = [&](int pair0, int n) {
    const unsigned pair_groups = kXcds * ((n + kXcds - 1) / kXcds);
    if (sweep) {
      const int s_max = std::max(w_left, w_top), l_max = std::max(H, W);
      const int n_col_chunks = (l_max / 8 + kThreads - 1) / kThreads;
      // rows per workgroup: kSweepRows when that still makes four rounds of workgroups over the chip, half as many for
      // a pass of few pairs (760 pairs: 0.69 against 0.75 ms with 64 rows; 95 pairs: 0.153 against 0.136 ms)
      const int64_t groups_at_most_rows = (int64_t)n * k * ((s_max + kSweepRows - 1) / kSweepRows + 2) * n_col_chunks;
      const int block_rows = groups_at_most_rows >= 4ll * MS_SWEEP_OCC * ctx->n_cus ? kSweepRows : std::max(1, kSweepRows / 2);
      const int n_row_chunks = (s_max + block_rows - 1) / block_rows + 2;
      MS_LAUNCH(ctx, "k_ncc_accumulate", k_ncc_sweep, dim3(pair_groups * k * n_row_chunks * n_col_chunks), dim3(kThreads), 0,
                ws.desc, status, pair0, n, ws.peaks, ws.cand, ws.acc, ws.n_cand, k, n_row_chunks, n_col_chunks, block_rows);
      if (subpixel)
        MS_LAUNCH(ctx, "k_ncc_accumulate", k_ncc_accumulate, dim3(pair_groups * kNccChunks), dim3(kThreads), 0, ws.desc, status,
                  pair0, n, ws.cand, ws.acc, ws.n_cand, ws.n_cand, 1);
    } else {
      MS_LAUNCH(ctx, "k_ncc_accumulate", k_ncc_accumulate, dim3(pair_groups * (ws.n_cand + 1) * kNccChunks), dim3(kThreads), 0,
                ws.desc, status, pair0, n, ws.cand, ws.acc, ws.n_cand, 0, ws.n_cand + 1);
    }
  };
  if (stage_all && (rc = stage(0, n_pairs))) return rc;
  // The block lists of every pass are merged into peaks and candidates by ONE launch behind the last pass when nothing in
  // between needs them (the subpixel stage does): a launch per pass is a dozen microseconds each for a few of work.
  const bool merge_at_end = stage_all && !subpixel;
  for (int pair0 = 0; pair0 < n_pairs; pair0 += chunk) {
    const int n = std::min(chunk, n_pairs - pair0);
    if (!stage_all && (rc = stage(pair0, n))) return rc;
    RegWorkspace wsp = ws;  // this pass's view: its pairs' block lists behind those of the passes before
    if (merge_at_end) wsp.partial = ws.partial + (int64_t)pair0 * ws.partial_stride;
    if (same_shape) {
      // left strips come from the aligned staging buffer, top strips from the tiles (or from staging too)
      if ((rc = launch_any(ctx, cfg_left, f64, wsp, status, pair0, n, -1, vec_strips, k, nullptr))) return rc;
    } else {
      if ((rc = launch_any(ctx, cfg_left, f64, wsp, status, pair0, n, 1, true, k, nullptr))) return rc;
      if ((rc = launch_any(ctx, cfg_top, f64, wsp, status, pair0, n, 0, vec_strips, k, nullptr))) return rc;
    }
    if (!merge_at_end)
      MS_LAUNCH(ctx, "k_peaks_candidates", k_peaks_candidates, dim3(n), dim3(kThreads), 0, ws.desc, status, pair0, ws.partial,
                ws.partial_stride, cfg_top.nblk, cfg_left.nblk, k, k, ws.peaks, ws.cand, ws.acc);
    if (subpixel)
      MS_LAUNCH(ctx, "k_ups_finish", k_ups_finish, dim3((n + 63) / 64), dim3(64), 0, ws.desc, status, pair0, n, ws.partial_ups,
                ws.n_row_blocks, ws.n_up, ws.uf, ws.ups, ws.cand, ws.n_cand);
    if (!stage_all) candidates(pair0, n);
  }
  if (merge_at_end)
    MS_LAUNCH(ctx, "k_peaks_candidates", k_peaks_candidates, dim3(n_pairs), dim3(kThreads), 0, ws.desc, status, 0, ws.partial,
              ws.partial_stride, cfg_top.nblk, cfg_left.nblk, k, k, ws.peaks, ws.cand, ws.acc);
  if (stage_all) candidates(0, n_pairs);
  MS_LAUNCH(ctx, "k_finalize", k_finalize, dim3((n_pairs + 3) / 4), dim3(256), 0, ws.desc, status, n_pairs, ws.cand, ws.acc,
                     out, status, subpixel ? ws.ups : (const UpsState*)nullptr, ws.n_cand, wide ? 1 : 0, table,
                     consensus ? 1 : 0);
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_register_pairs(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h,
                                       int tile_w, int64_t tile_pitch, const int32_t* pair_idx,
                                       const uint8_t* pair_dir, int n_pairs, int edge_width, unsigned flags,
                                       double* out, int8_t* status) {
  return register_pairs_impl(ctx, tiles, n_tiles, tile_h, tile_w, tile_pitch, nullptr, MISTITCH_FF_NONE, pair_idx, pair_dir,
                             n_pairs, edge_width, flags, out, status);
}

extern "C" int mistitch_register_pairs_flatfield(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h,
                                                 int tile_w, int64_t tile_pitch, const void* flatfield, int ff_type,
                                                 const int32_t* pair_idx, const uint8_t* pair_dir, int n_pairs,
                                                 int edge_width, unsigned flags, double* out, int8_t* status) {
  MS_REQUIRE(flatfield, MISTITCH_ERR_ARG, "flatfield is NULL");
  MS_REQUIRE(ff_type == MISTITCH_FF_F32 || ff_type == MISTITCH_FF_F64, MISTITCH_ERR_ARG, "unknown flatfield type %d",
             ff_type);
  return register_pairs_impl(ctx, tiles, n_tiles, tile_h, tile_w, tile_pitch, flatfield, ff_type, pair_idx, pair_dir,
                             n_pairs, edge_width, flags, out, status);
}

extern "C" int mistitch_register_pairs_candidates(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h,
                                                  int tile_w, int64_t tile_pitch, const void* flatfield, int ff_type,
                                                  const int32_t* pair_idx, const uint8_t* pair_dir, int n_pairs,
                                                  int edge_width, unsigned flags, double* out, int8_t* status,
                                                  double* table) {
  MS_REQUIRE(table, MISTITCH_ERR_ARG, "table is NULL");
  MS_REQUIRE(ff_type == MISTITCH_FF_NONE || ff_type == MISTITCH_FF_F32 || ff_type == MISTITCH_FF_F64, MISTITCH_ERR_ARG,
             "unknown flatfield type %d", ff_type);
  MS_REQUIRE((ff_type == MISTITCH_FF_NONE) == (flatfield == nullptr), MISTITCH_ERR_ARG,
             "flatfield and its type disagree");
  return register_pairs_impl(ctx, tiles, n_tiles, tile_h, tile_w, tile_pitch, flatfield, ff_type, pair_idx, pair_dir,
                             n_pairs, edge_width, flags, out, status, table);
}

// Shared body of mistitch_pcm / mistitch_pcm_topk: one pair of free-standing strips.
__global__ void k_ups_to_array(const PairDesc* desc, const UpsState* ups, double* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const PairDesc d = desc[0];
  const UpsState u = ups[0];
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  out[0] = u.valid ? (d.transposed ? u.sub_l : u.sub_s) : nan;
  out[1] = u.valid ? (d.transposed ? u.sub_s : u.sub_l) : nan;
}

static int single_pair_pcm(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1, const uint16_t* strip2,
                           int64_t pitch2, int h, int w, unsigned flags, int k, double* out_pcm, int32_t* rows,
                           int32_t* cols, double* vals, int upsample = 0, double* sub_out = nullptr) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(strip1 && strip2, MISTITCH_ERR_ARG, "NULL strip");
  MS_REQUIRE(h > 0 && w > 0 && pitch1 >= w && pitch2 >= w, MISTITCH_ERR_ARG, "bad strip geometry");
  MS_REQUIRE(k >= 0 && k <= kMaxK, MISTITCH_ERR_ARG, "k must be in [0, %d]", kMaxK);
  MS_HIP(hipSetDevice(ctx->device));
  const bool f64 = !(flags & MISTITCH_FFT_F32);
  const bool transposed = h > w;  // tall strips are processed transposed
  const int S = transposed ? w : h, L = transposed ? h : w;
  AnyShape cfg;
  int rc;
  if ((rc = make_any_shape(ctx, S, L, f64, &cfg))) return rc;
  const int pstride = cfg.nblk * kMaxK;
  RegWorkspace ws;
  // the staging slot holds BOTH strips (transposed, or re-pitched) one after the other
  if ((rc = reserve_registration(ctx, 1, 1, 1, (size_t)2 * S * L, cfg.spec_bytes, cfg.slab_bytes, pstride, &ws))) return rc;
  ws.window = (flags & MISTITCH_REG_WINDOW_HANN) != 0;
  if (sub_out != nullptr) {
    if ((rc = reserve_subpixel(ctx, 1, 1, cfg.nblk, cfg.n_stored, upsample, &ws))) return rc;
  }
  if ((rc = ctx->misc.reserve(256))) return rc;
  hipStream_t st = ctx->stream;
  PairDesc d;
  bool vec = true;
  if (transposed) {
    MS_LAUNCH(ctx, "k_transpose_plain", k_transpose_plain, dim3((h + 31) / 32, (w + 31) / 32), dim3(32, 8), 0, strip1, pitch1, h, w,
                       ws.staging);
    MS_LAUNCH(ctx, "k_transpose_plain", k_transpose_plain, dim3((h + 31) / 32, (w + 31) / 32), dim3(32, 8), 0, strip2, pitch2, h, w,
                       ws.staging + (size_t)S * L);
    d.a = ws.staging;
    d.b = ws.staging + (size_t)S * L;
    d.pitch = L;
  } else {
    // the kernels share one pitch per pair; a strip with another pitch is staged
    d.a = strip1;
    d.b = strip2;
    d.pitch = pitch1;
    if (pitch2 != pitch1) {
      MS_HIP(hipMemcpy2DAsync(ws.staging, (size_t)L * 2, strip1, (size_t)pitch1 * 2, (size_t)L * 2, S,
                              hipMemcpyDeviceToDevice, st));
      MS_HIP(hipMemcpy2DAsync(ws.staging + (size_t)S * L, (size_t)L * 2, strip2, (size_t)pitch2 * 2, (size_t)L * 2, S,
                              hipMemcpyDeviceToDevice, st));
      d.a = ws.staging;
      d.b = ws.staging + (size_t)S * L;
      d.pitch = L;
    }
    vec = aligned16(d.a) && aligned16(d.b) && (d.pitch % 8 == 0);
  }
  d.S = S;
  d.L = L;
  d.transposed = transposed ? 1 : 0;
  d.img_h = h;
  d.img_w = w;
  d.off_y = d.off_x = 0;
  int8_t* status = reinterpret_cast<int8_t*>(ctx->misc.ptr);
  MS_HIP(hipMemsetAsync(status, 0, 16, st));
  MS_HIP(hipMemcpyAsync(ws.desc, &d, sizeof(d), hipMemcpyHostToDevice, st));
  MS_HIP(hipStreamSynchronize(st));  // d lives on this stack frame
  if ((rc = launch_any(ctx, cfg, f64, ws, status, 0, 1, -1, vec, k, out_pcm))) return rc;
  if (sub_out != nullptr) {
    MS_LAUNCH(ctx, "k_ups_finish", k_ups_finish, dim3(1), dim3(64), 0, ws.desc, status, 0, 1, ws.partial_ups, ws.n_row_blocks,
              ws.n_up, ws.uf, ws.ups, ws.cand, ws.n_cand);
    MS_LAUNCH(ctx, "k_ups_to_array", k_ups_to_array, dim3(1), dim3(64), 0, ws.desc, ws.ups, sub_out);
    MS_HIP(hipGetLastError());
    return MISTITCH_OK;
  }
  if (k > 0) {
    MS_LAUNCH(ctx, "k_peaks_candidates", k_peaks_candidates, dim3(1), dim3(kThreads), 0, ws.desc, status, 0, ws.partial,
                       ws.partial_stride, cfg.nblk, cfg.nblk, k, 0, ws.peaks,
                       (Cand*)nullptr, (unsigned long long*)nullptr);
    MS_LAUNCH(ctx, "k_peaks_to_arrays", k_peaks_to_arrays, dim3(1), dim3(64), 0, ws.peaks, k, w, rows, cols, vals);
  }
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_pcm(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1, const uint16_t* strip2,
                            int64_t pitch2, int h, int w, unsigned flags, double* out_pcm) {
  MS_REQUIRE(out_pcm, MISTITCH_ERR_ARG, "out_pcm is NULL");
  return single_pair_pcm(ctx, strip1, pitch1, strip2, pitch2, h, w, flags, 0, out_pcm, nullptr, nullptr, nullptr);
}

extern "C" int mistitch_pcm_topk(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1, const uint16_t* strip2,
                                 int64_t pitch2, int h, int w, unsigned flags, int k, int32_t* rows, int32_t* cols,
                                 double* vals) {
  MS_REQUIRE(rows && cols && vals, MISTITCH_ERR_ARG, "NULL output array");
  MS_REQUIRE(k >= 1 && k <= kMaxK, MISTITCH_ERR_ARG, "k must be in [1, %d]", kMaxK);
  return single_pair_pcm(ctx, strip1, pitch1, strip2, pitch2, h, w, flags, k, nullptr, rows, cols, vals);
}

extern "C" int mistitch_refine_pairs(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                                     int64_t tile_pitch, const int32_t* pair_idx, const uint8_t* pair_dir, int n_pairs,
                                     int edge_width, const int32_t* start_yx, double r, double* out, int8_t* status) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(n_pairs >= 0 && n_tiles >= 0, MISTITCH_ERR_ARG, "negative count");
  if (n_pairs == 0) return MISTITCH_OK;
  MS_REQUIRE(tiles && pair_idx && pair_dir && start_yx && out && status, MISTITCH_ERR_ARG, "NULL array argument");
  MS_REQUIRE(tile_h > 0 && tile_w > 0 && edge_width > 0, MISTITCH_ERR_ARG, "non-positive size");
  MS_REQUIRE(tile_pitch >= (int64_t)tile_h * tile_w, MISTITCH_ERR_ARG, "tile_pitch smaller than a tile");
  MS_REQUIRE(r >= 0.0, MISTITCH_ERR_ARG, "r must be non-negative");
  MS_HIP(hipSetDevice(ctx->device));
  const int H = tile_h, W = tile_w;
  const int w_left = edge_width < W ? edge_width : W;
  const int w_top = edge_width < H ? edge_width : H;
  const int chunk = n_pairs < ctx->pair_chunk ? n_pairs : ctx->pair_chunk;
  int rc;
  if ((rc = ctx->desc.reserve((size_t)n_pairs * sizeof(PairDesc)))) return rc;
  if ((rc = ctx->staging.reserve((size_t)chunk * w_left * H * 2 * 2 + 256))) return rc;
  PairDesc* desc = (PairDesc*)ctx->desc.ptr;
  uint16_t* staging = (uint16_t*)ctx->staging.ptr;
  const bool vec_top = aligned16(tiles) && (W % 8 == 0) && (tile_pitch % 8 == 0);
  MS_LAUNCH(ctx, "k_setup_pairs", k_setup_pairs, dim3((n_pairs + 255) / 256), dim3(256), 0, tiles, tile_pitch, H, W, w_left,
            w_top, pair_idx, pair_dir, n_pairs, n_tiles, staging, chunk, desc, status, (int64_t)2 * w_left * H, 0);
  for (int pair0 = 0; pair0 < n_pairs; pair0 += chunk) {
    const int n = std::min(chunk, n_pairs - pair0);
    if (vec_top && H % 64 == 0 && w_left % 64 == 0)
      MS_LAUNCH(ctx, "k_transpose_strips", k_transpose_strips64<false>, dim3(H / 64, w_left / 64, 2 * n), dim3(256), 0, tiles,
                tile_pitch, H, W, w_left, pair_idx, desc, status, pair0);
    else
      MS_LAUNCH(ctx, "k_transpose_strips", k_transpose_strips, dim3((H + 31) / 32, (w_left + 31) / 32, 2 * n), dim3(32, 8), 0,
                tiles, tile_pitch, H, W, w_left, pair_idx, desc, status, pair0);
    MS_LAUNCH(ctx, "k_refine_pairs", k_refine_pairs, dim3(n), dim3(kThreads), 0, desc, status, pair0, start_yx, r, H, W, out);
  }
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_subpixel_shift(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1, const uint16_t* strip2,
                                       int64_t pitch2, int h, int w, unsigned flags, int upsample_factor,
                                       double* out_shift) {
  MS_REQUIRE(out_shift, MISTITCH_ERR_ARG, "out_shift is NULL");
  return single_pair_pcm(ctx, strip1, pitch1, strip2, pitch2, h, w, flags, 2, nullptr, nullptr, nullptr, nullptr,
                         upsample_factor, out_shift);
}

extern "C" int mistitch_set_upsample_factor(mistitch_ctx* ctx, int upsample_factor) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(upsample_factor >= 2 && (3 * upsample_factor + 1) / 2 <= kUpsMax, MISTITCH_ERR_UNSUPPORTED,
             "upsample factor %d outside [2, %d]", upsample_factor, 2 * kUpsMax / 3);
  ctx->upsample_factor = upsample_factor;
  return MISTITCH_OK;
}

extern "C" int mistitch_ncc(mistitch_ctx* ctx, const uint16_t* patch1, int64_t pitch1, const uint16_t* patch2,
                            int64_t pitch2, int h, int w, int min_overlap_pixels, double* out) {
  MS_REQUIRE(ctx && out, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(h >= 0 && w >= 0, MISTITCH_ERR_ARG, "negative size");
  MS_REQUIRE((patch1 && patch2) || h * (long long)w == 0, MISTITCH_ERR_ARG, "NULL patch");
  MS_REQUIRE((long long)h * w <= (1ll << 30), MISTITCH_ERR_UNSUPPORTED, "patch larger than 2^30 pixels");
  MS_HIP(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ctx->misc.reserve(256))) return rc;
  hipStream_t st = ctx->stream;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->misc.ptr) + 64);
  MS_HIP(hipMemsetAsync(acc, 0, 5 * sizeof(unsigned long long), st));
  if ((long long)h * w > 0)
    MS_LAUNCH(ctx, "k_ncc_single", k_ncc_single, dim3(h < 1024 ? h : 1024), dim3(kThreads), 0, patch1, pitch1, patch2, pitch2,
                       h, w, acc);
  MS_LAUNCH(ctx, "k_ncc_single_finish", k_ncc_single_finish, dim3(1), dim3(64), 0, acc, (long long)h * w,
            (long long)min_overlap_pixels, out);
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}

extern "C" int mistitch_edge_variance(mistitch_ctx* ctx, const uint16_t* strip, int64_t pitch, int h, int w,
                                      double* out_var) {
  MS_REQUIRE(ctx && strip && out_var, MISTITCH_ERR_ARG, "NULL argument");
  MS_REQUIRE(h > 0 && w > 0 && pitch >= w, MISTITCH_ERR_ARG, "bad strip geometry");
  MS_HIP(hipSetDevice(ctx->device));
  MS_LAUNCH(ctx, "k_variance_single", k_variance_single, dim3(1), dim3(kThreads), 0, strip, pitch, h, w, out_var);
  MS_HIP(hipGetLastError());
  return MISTITCH_OK;
}
