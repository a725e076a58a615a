// Ceiling of a TILE-ROW-centric fusion pass on the benchmark geometry (20 x 20 x 4 tiles of
// 2048^2 on a 1843-pixel pitch with jitter): a wave keeps one flatfield row in registers and
// walks the 20 tiles of a grid row with it, reading each tile row with aligned 16-byte loads,
// dividing in the tile's own alignment, shifting the corrected pixels to the mosaic's 16-byte grid
// through DPP + funnel shifts and writing aligned 16-byte groups.  No ownership test: every tile
// pixel is written (overlaps twice), so the bytes are 2 B read + 2 B written per TILE pixel.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/tilecopy_probe.hip -o tools/tilecopy_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int T = 2048, GRID = 20, CH = 4, GX = 4;

template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ f2 div2(f2 n, f2 d) {
  f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 one = {1.0f, 1.0f};
  const f2 e = __builtin_elementwise_fma(-d, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  const f2 q = n * r;
  const f2 t = __builtin_elementwise_fma(-d, q, n);
  return __builtin_elementwise_fma(t, r, q);
}

__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_u16(lo, hi));
}

// grid: CH * (T / RPW / 4) * GRID workgroups; wave w of block b: rows [RPW * (4 b' + w), +RPW) of every tile of grid row gr.
template <bool FF, int RPW, int ORDER = 0, int PF = 0>
__global__ __launch_bounds__(256) void k_tile_rows(const uint16_t* __restrict__ tiles, const float* __restrict__ ffs,
                                                   const int* __restrict__ yx, uint16_t* __restrict__ mosaic, int Hm, int Wm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = blockIdx.x;
  int gr, rb, p;
  if constexpr (ORDER == 0) {
    gr = b % GRID;
    b /= GRID;
    rb = b % (T / RPW / 4);
    p = b / (T / RPW / 4);
  } else {
    rb = b % (T / RPW / 4);
    b /= (T / RPW / 4);
    gr = b % GRID;
    p = b / GRID;
  }
  for (int rr = 0; rr < RPW; ++rr) {
    const int ly = RPW * (4 * rb + wave) + rr;
    f4 ff[GX][2];
    if constexpr (FF) {
#pragma unroll
      for (int r = 0; r < GX; ++r) {
        const float* f = ffs + ((size_t)p * T + ly) * T + 8 * (lane + 64 * r);
        ff[r][0] = *reinterpret_cast<const f4*>(f);
        ff[r][1] = *reinterpret_cast<const f4*>(f + 4);
      }
    }
    u4 nxt[GX];
    if constexpr (PF) {
      const uint16_t* src0 = tiles + ((size_t)((p * GRID + gr) * GRID) * T + ly) * T;
#pragma unroll
      for (int r = 0; r < GX; ++r) nxt[r] = *reinterpret_cast<const u4*>(src0 + 8 * (lane + 64 * r));
    }
    for (int c = 0; c < GRID; ++c) {
      const int t = (p * GRID + gr) * GRID + c;
      const int ty = yx[2 * t], tx = yx[2 * t + 1];
      const int y = ty + ly;
      const size_t row_el = ((size_t)p * Hm + y) * Wm;
      const int a = (int)(row_el & 7);
      uint16_t* rowa = mosaic + (row_el - a);  // 16-byte aligned
      const int g0 = (tx + a) >> 3;
      const int k = tx + a - 8 * g0;  // 0..7: output group j starts k pixels before source group j
      const uint16_t* src = tiles + ((size_t)t * T + ly) * T;
      u4 v[GX];
      if constexpr (PF) {
#pragma unroll
        for (int r = 0; r < GX; ++r) v[r] = nxt[r];
        if (c + 1 < GRID) {
#pragma unroll
          for (int r = 0; r < GX; ++r) nxt[r] = *reinterpret_cast<const u4*>(src + (size_t)T * T + 8 * (lane + 64 * r));
        }
      } else {
#pragma unroll
        for (int r = 0; r < GX; ++r) v[r] = *reinterpret_cast<const u4*>(src + 8 * (lane + 64 * r));
      }
      if constexpr (FF) {
#pragma unroll
        for (int r = 0; r < GX; ++r) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned w = v[r][d];
            const f2 n = {(float)(w & 0xffffu), (float)(w >> 16)};
            const f2 den = {ff[r][d >> 1][2 * (d & 1)], ff[r][d >> 1][2 * (d & 1) + 1]};
            const f2 q = div2(n, den);
            v[r][d] = pack2((unsigned)q.x, (unsigned)q.y);
          }
        }
      }
      unsigned R[GX][4];
#pragma unroll
      for (int r = 0; r < GX; ++r)
#pragma unroll
        for (int d = 0; d < 4; ++d) R[r][d] = dpp<0x13c>(v[r][d]);  // wave_ror:1: lane l reads lane l - 1, lane 0 reads lane 63
      const unsigned p0 = 8 - k, q = p0 >> 1, hb = (p0 & 1) * 16;
#pragma unroll
      for (int r = 0; r < GX; ++r) {
        unsigned C[9];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned before = r == 0 ? 0u : R[r - 1][d];
          C[d] = lane == 0 ? before : R[r][d];
          C[4 + d] = v[r][d];
        }
        C[8] = 0;
        u4 o;
        switch (q) {  // wave-uniform
          case 0:
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 1], C[d], hb);
            break;
          case 1:
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 2], C[d + 1], hb);
            break;
          case 2:
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 3], C[d + 2], hb);
            break;
          case 3:
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 4], C[d + 3], hb);
            break;
          default:
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = C[d + 4];
            break;
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u4*>(rowa + 8 * (size_t)(g0 + lane + 64 * r)));
      }
    }
  }
}

// One group per lane: a workgroup of 256 lanes takes ONE tile row (4 KiB); LOOP = 1: and walks the 20 tiles
// of the grid row with the flatfield values of its 8 pixels in registers; LOOP = 0: one tile, flatfield from memory.
// FFMODE (timing only, the values then belong to other pixels): 1 = the flatfield loads of a wave are contiguous
// (16 bytes per lane at a 16-byte stride, two halves of the row) instead of 16 bytes at a 32-byte stride;
// 2 = non-temporal flatfield loads; 3 = one 16-byte load only (half the flatfield bytes).
template <bool FF, int LOOP, int PF, int FFMODE = 0>
__global__ __launch_bounds__(256) void k_tile_row1(const uint16_t* __restrict__ tiles, const float* __restrict__ ffs,
                                                   const int* __restrict__ yx, uint16_t* __restrict__ mosaic, int Hm, int Wm) {
  const int lane = threadIdx.x & 63, j = threadIdx.x;
  int b = blockIdx.x;
  const int ly = b % T;
  b /= T;
  int t0, nt;
  if constexpr (LOOP) {
    t0 = b * GRID;  // b = p * GRID + gr
    nt = GRID;
  } else {
    t0 = b;
    nt = 1;
  }
  const int p = t0 / (GRID * GRID);
  f4 ff[2];
  if constexpr (FF) {
    const float* f = ffs + ((size_t)p * T + ly) * T + 8 * j;
    if constexpr (FFMODE == 1) {
      const float* g = ffs + ((size_t)p * T + ly) * T + 4 * j;
      ff[0] = *reinterpret_cast<const f4*>(g);
      ff[1] = *reinterpret_cast<const f4*>(g + T / 2);
    } else if constexpr (FFMODE == 2) {
      ff[0] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(f));
      ff[1] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(f + 4));
    } else if constexpr (FFMODE == 3) {
      ff[0] = *reinterpret_cast<const f4*>(f);
      ff[1] = ff[0];
    } else if constexpr (FFMODE == 4) {  // every flatfield load falls into one 8 KiB row: served by the CU's L1
      const float* g = ffs + 8 * j;
      ff[0] = *reinterpret_cast<const f4*>(g);
      ff[1] = *reinterpret_cast<const f4*>(g + 4);
    } else if constexpr (FFMODE == 5) {  // a 512 KiB window (64 rows): served by L2, never by L1
      const float* g = ffs + (size_t)(ly & 63) * T + 8 * j;
      ff[0] = *reinterpret_cast<const f4*>(g);
      ff[1] = *reinterpret_cast<const f4*>(g + 4);
    } else {
      ff[0] = *reinterpret_cast<const f4*>(f);
      ff[1] = *reinterpret_cast<const f4*>(f + 4);
    }
  }
  u4 nxt, nxt_prev;
  if constexpr (PF) {
    const uint16_t* src0 = tiles + ((size_t)t0 * T + ly) * T;
    nxt = *reinterpret_cast<const u4*>(src0 + 8 * j);
    nxt_prev = u4{0, 0, 0, 0};
    if (lane == 0 && j > 0) nxt_prev = *reinterpret_cast<const u4*>(src0 + 8 * j - 8);
  }
  for (int c = 0; c < nt; ++c) {
    const int t = t0 + c;
    const int ty = yx[2 * t], tx = yx[2 * t + 1];
    const int y = ty + ly;
    const size_t row_el = ((size_t)p * Hm + y) * Wm;
    const int a = (int)(row_el & 7);
    uint16_t* rowa = mosaic + (row_el - a);
    const int g0 = (tx + a) >> 3;
    const int k = tx + a - 8 * g0;
    const uint16_t* src = tiles + ((size_t)t * T + ly) * T;
    u4 v, pv = u4{0, 0, 0, 0};
    if constexpr (PF) {
      v = nxt;
      pv = nxt_prev;
      if (c + 1 < nt) {
        nxt = *reinterpret_cast<const u4*>(src + (size_t)T * T + 8 * j);
        if (lane == 0 && j > 0) nxt_prev = *reinterpret_cast<const u4*>(src + (size_t)T * T + 8 * j - 8);
      }
    } else {
      v = *reinterpret_cast<const u4*>(src + 8 * j);
      if (lane == 0 && j > 0) pv = *reinterpret_cast<const u4*>(src + 8 * j - 8);  // the group before this wave's first
    }
    auto correct = [&](u4& x) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w = x[d];
        const f2 n = {(float)(w & 0xffffu), (float)(w >> 16)};
        const f2 den = {ff[d >> 1][2 * (d & 1)], ff[d >> 1][2 * (d & 1) + 1]};
        const f2 q = div2(n, den);
        x[d] = pack2((unsigned)q.x, (unsigned)q.y);
      }
    };
    if constexpr (FF) correct(v);
    // (the probe leaves lane 0's extra group uncorrected: it needs the previous group's flatfield values)
    unsigned C[9];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned r = dpp<0x138>(v[d]);  // wave_shr:1
      C[d] = lane == 0 ? pv[d] : r;
      C[4 + d] = v[d];
    }
    C[8] = 0;
    const unsigned p0 = 8 - k, q = p0 >> 1, hb = (p0 & 1) * 16;
    u4 o;
    switch (q) {
      case 0:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 1], C[d], hb);
        break;
      case 1:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 2], C[d + 1], hb);
        break;
      case 2:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 3], C[d + 2], hb);
        break;
      case 3:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 4], C[d + 3], hb);
        break;
      default:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = C[d + 4];
        break;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<u4*>(rowa + 8 * (size_t)(g0 + j)));
  }
}

// One group per lane and tile; a workgroup takes row ly of NT consecutive tiles (all their loads issued
// together, then corrected with the SAME flatfield registers, shifted and stored).
template <bool FF, int NT, bool NTL = false, int ORD = 0>
__global__ __launch_bounds__(256) void k_tile_rowN(const uint16_t* __restrict__ tiles, const float* __restrict__ ffs,
                                                   const int* __restrict__ yx, uint16_t* __restrict__ mosaic, int Hm, int Wm) {
  const int lane = threadIdx.x & 63, j = threadIdx.x;
  int b = blockIdx.x;
  int ly, t0;
  if constexpr (ORD == 0) {  // tile by tile: rows of one tile are taken by consecutive workgroups
    ly = b % T;
    b /= T;
    t0 = b * NT;
  } else {  // mosaic order: the tiles of a grid row fastest, then the tile row, then the grid row
    const int per_row = GRID / NT;
    const int c = b % per_row;
    b /= per_row;
    if constexpr (ORD == 1) {
      ly = b % T;
      b /= T;  // = p * GRID + gr
    } else {   // ORD 2: 8 XCD-interleaved: rows in bunches so that one XCD sees consecutive rows
      ly = b % T;
      b /= T;
    }
    t0 = b * GRID + c * NT;
  }
  const int p = t0 / (GRID * GRID);
  f4 ff[2];
  u4 v[NT], pv[NT];
#pragma unroll
  for (int c = 0; c < NT; ++c) {
    const uint16_t* src = tiles + ((size_t)(t0 + c) * T + ly) * T;
    v[c] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + 8 * j)) : *reinterpret_cast<const u4*>(src + 8 * j);
    pv[c] = u4{0, 0, 0, 0};
    if (lane == 0 && j > 0) pv[c] = *reinterpret_cast<const u4*>(src + 8 * j - 8);
  }
  if constexpr (FF) {
    const float* f = ffs + ((size_t)p * T + ly) * T + 8 * j;
    ff[0] = *reinterpret_cast<const f4*>(f);
    ff[1] = *reinterpret_cast<const f4*>(f + 4);
  }
#pragma unroll
  for (int c = 0; c < NT; ++c) {
    const int t = t0 + c;
    const int ty = yx[2 * t], tx = yx[2 * t + 1];
    const int y = ty + ly;
    const size_t row_el = ((size_t)p * Hm + y) * Wm;
    const int a = (int)(row_el & 7);
    uint16_t* rowa = mosaic + (row_el - a);
    const int g0 = (tx + a) >> 3;
    const int k = tx + a - 8 * g0;
    if constexpr (FF) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w = v[c][d];
        const f2 n = {(float)(w & 0xffffu), (float)(w >> 16)};
        const f2 den = {ff[d >> 1][2 * (d & 1)], ff[d >> 1][2 * (d & 1) + 1]};
        const f2 q = div2(n, den);
        v[c][d] = pack2((unsigned)q.x, (unsigned)q.y);
      }
    }
    unsigned C[9];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned r = dpp<0x138>(v[c][d]);
      C[d] = lane == 0 ? pv[c][d] : r;
      C[4 + d] = v[c][d];
    }
    C[8] = 0;
    const unsigned p0 = 8 - k, q = p0 >> 1, hb = (p0 & 1) * 16;
    u4 o;
    switch (q) {
      case 0:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 1], C[d], hb);
        break;
      case 1:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 2], C[d + 1], hb);
        break;
      case 2:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 3], C[d + 2], hb);
        break;
      case 3:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_alignbit(C[d + 4], C[d + 3], hb);
        break;
      default:
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = C[d + 4];
        break;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<u4*>(rowa + 8 * (size_t)(g0 + j)));
  }
}

int main() {
  const int pitch = T - 205;
  std::vector<int> yx(2 * CH * GRID * GRID);
  unsigned seed = 12345;
  auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return (int)((seed >> 16) % 7); };
  for (int p = 0; p < CH; ++p)
    for (int r = 0; r < GRID; ++r)
      for (int c = 0; c < GRID; ++c) {
        const int t = (p * GRID + r) * GRID + c;
        yx[2 * t] = r * pitch + rnd();
        yx[2 * t + 1] = c * pitch + rnd();
      }
  const int Hm = (GRID - 1) * pitch + 6 + T, Wm = (GRID - 1) * pitch + 6 + T + 1;
  const size_t n_tiles = (size_t)CH * GRID * GRID;
  uint16_t *tiles, *mosaic;
  float* ffs;
  int* dyx;
  const size_t tile_bytes = n_tiles * T * T * 2, mosaic_bytes = (size_t)CH * Hm * Wm * 2 + 65536;
  CK(hipMalloc(&tiles, tile_bytes));
  CK(hipMalloc(&mosaic, mosaic_bytes));
  CK(hipMalloc(&ffs, (size_t)CH * T * T * 4));
  CK(hipMalloc(&dyx, yx.size() * 4));
  CK(hipMemset(tiles, 0x11, tile_bytes));
  std::vector<float> hff((size_t)CH * T * T);
  for (size_t i = 0; i < hff.size(); ++i) hff[i] = 0.8f + 0.4f * (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
  CK(hipMemcpy(ffs, hff.data(), hff.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dyx, yx.data(), yx.size() * 4, hipMemcpyHostToDevice));
  printf("mosaic %d x %d x %d, %zu tiles: %.2f GB read + %.2f GB written per pass\n", CH, Hm, Wm, n_tiles, tile_bytes / 1e9,
         tile_bytes / 1e9);
  auto run = [&](const char* name, auto launch) {
    launch();
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int reps = 5;
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-40s %7.3f ms  %7.1f GB/s\n", name, ms / reps, 2.0 * tile_bytes / (ms / reps * 1e-3) / 1e9);
    fflush(stdout);
  };
  run("tile rows, no flatfield, 1 row/wave", [&] { k_tile_rows<false, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("tile rows, flatfield f32, 1 row/wave", [&] { k_tile_rows<true, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("tile rows, no flatfield, 2 rows/wave", [&] { k_tile_rows<false, 2><<<CH * (T / 8) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("tile rows, flatfield f32, 2 rows/wave", [&] { k_tile_rows<true, 2><<<CH * (T / 8) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("ff, 1 row/wave, row band fastest", [&] { k_tile_rows<true, 1, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("no ff, 1 row/wave, row band fastest", [&] { k_tile_rows<false, 1, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("ff, 1 row/wave, prefetch next tile", [&] { k_tile_rows<true, 1, 0, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("ff, 1 row/wave, band fastest + prefetch", [&] { k_tile_rows<true, 1, 1, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("no ff, band fastest + prefetch", [&] { k_tile_rows<false, 1, 1, 1><<<CH * (T / 4) * GRID, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  const unsigned n_rows_all = (unsigned)(n_tiles * T), n_rows_loop = (unsigned)(CH * GRID * T);
  run("1 group/lane, one tile row per WG, no ff", [&] { k_tile_row1<false, 0, 0><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("1 group/lane, one tile row per WG, ff mem", [&] { k_tile_row1<true, 0, 0><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("  the same, contiguous ff loads (timing)", [&] { k_tile_row1<true, 0, 0, 1><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("  the same, non-temporal ff loads", [&] { k_tile_row1<true, 0, 0, 2><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("  the same, ff loads hit L1 (timing)", [&] { k_tile_row1<true, 0, 0, 4><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("  the same, ff loads hit L2 only (timing)", [&] { k_tile_row1<true, 0, 0, 5><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("  the same, half the ff bytes (timing)", [&] { k_tile_row1<true, 0, 0, 3><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("1 group/lane, one tile row per WG, no ff", [&] { k_tile_row1<false, 0, 0><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("1 group/lane, one tile row per WG, ff mem", [&] { k_tile_row1<true, 0, 0><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("1 group/lane, 20 tiles per WG, no ff", [&] { k_tile_row1<false, 1, 0><<<n_rows_loop, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("1 group/lane, 20 tiles per WG, ff regs", [&] { k_tile_row1<true, 1, 0><<<n_rows_loop, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("1 group/lane, 20 tiles, ff regs, prefetch", [&] { k_tile_row1<true, 1, 1><<<n_rows_loop, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 1 tile per WG, no ff", [&] { k_tile_rowN<false, 1><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 1 tile per WG, ff", [&] { k_tile_rowN<true, 1><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 2 tiles per WG, no ff", [&] { k_tile_rowN<false, 2><<<n_rows_all / 2, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 2 tiles per WG, ff", [&] { k_tile_rowN<true, 2><<<n_rows_all / 2, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 4 tiles per WG, no ff", [&] { k_tile_rowN<false, 4><<<n_rows_all / 4, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 4 tiles per WG, ff", [&] { k_tile_rowN<true, 4><<<n_rows_all / 4, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN: 5 tiles per WG, ff", [&] { k_tile_rowN<true, 5><<<n_rows_all / 5, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN nt loads: 1 tile per WG, no ff", [&] { k_tile_rowN<false, 1, true><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN nt loads: 1 tile per WG, ff", [&] { k_tile_rowN<true, 1, true><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN nt loads: 2 tiles per WG, no ff", [&] { k_tile_rowN<false, 2, true><<<n_rows_all / 2, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN nt loads: 2 tiles per WG, ff", [&] { k_tile_rowN<true, 2, true><<<n_rows_all / 2, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN nt loads: 4 tiles per WG, ff", [&] { k_tile_rowN<true, 4, true><<<n_rows_all / 4, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN mosaic order: 1 tile per WG, no ff", [&] { k_tile_rowN<false, 1, false, 1><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN mosaic order: 1 tile per WG, ff", [&] { k_tile_rowN<true, 1, false, 1><<<n_rows_all, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN mosaic order: 2 tiles per WG, ff", [&] { k_tile_rowN<true, 2, false, 1><<<n_rows_all / 2, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN mosaic order nt: 2 tiles per WG, ff", [&] { k_tile_rowN<true, 2, true, 1><<<n_rows_all / 2, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  run("rowN mosaic order: 4 tiles per WG, ff", [&] { k_tile_rowN<true, 4, false, 1><<<n_rows_all / 4, 256>>>(tiles, ffs, dyx, mosaic, Hm, Wm); });
  return 0;
}
