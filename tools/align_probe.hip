// What misaligned 16-byte loads cost in a streaming copy, and what the alternatives cost:
// rows of uint16 pixels copied from a source whose rows start `shift` pixels off the 16-byte
// grid into 16-byte aligned destination groups.  Variants:
//   unaligned   one 16-byte load per lane at the misaligned address (2-byte aligned)
//   dpp         aligned 16-byte loads, the neighbouring lane's dwords by DPP wave_shl:1, funnel shift
//   two_loads   two aligned 16-byte loads per lane (groups j and j + 1), funnel shift, no cross-lane step
//   dword5      one dword-aligned 16-byte load plus one dword, funnel shift by a halfword
// Build: hipcc -O3 --offload-arch=gfx950 tools/align_probe.hip -o tools/align_probe
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
struct __attribute__((packed, aligned(2))) P8 {
  unsigned d[4];
};
struct __attribute__((packed, aligned(4))) Q8 {
  unsigned d[4];
};

constexpr int GX = 4;  // groups per lane and row, 64 groups apart: a wave moves 4 KiB of one row

template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, false);
}

__global__ void k_dpp_semantics(unsigned* out) {
  const unsigned lane = threadIdx.x;
  out[lane] = dpp<0x130>(lane);        // wave_shl:1
  out[64 + lane] = dpp<0x134>(lane);   // wave_rol:1
  out[128 + lane] = dpp<0x138>(lane);  // wave_shr:1
  out[192 + lane] = dpp<0x13c>(lane);  // wave_ror:1
}

// dst row r, group g (aligned) <- src pixels [8 g + shift, 8 g + shift + 8) of src row r.
// grid: (row blocks); a workgroup of 256 lanes takes rows wave by wave.
template <int MODE>
__global__ __launch_bounds__(256) void k_copy_rows(const uint16_t* __restrict__ src, int64_t src_pitch, uint16_t* __restrict__ dst,
                                                   int64_t dst_pitch, int n_rows, int groups_per_row, int shift) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = groups_per_row / (64 * GX);
  for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < (int64_t)n_rows * chunks; item += (int64_t)gridDim.x * 4) {
    const int r = (int)(item / chunks), c = (int)(item % chunks);
    const uint16_t* s = src + r * src_pitch + (int64_t)c * (64 * GX * 8);
    uint16_t* d = dst + r * dst_pitch + (int64_t)c * (64 * GX * 8);
    // the shift of a row differs from row to row as it does in a mosaic with odd row length
    const int sh = (shift + r) & 7;
    if constexpr (MODE == 0) {  // unaligned
      P8 v[GX];
#pragma unroll
      for (int i = 0; i < GX; ++i) v[i] = *reinterpret_cast<const P8*>(s + sh + 8 * (lane + 64 * i));
#pragma unroll
      for (int i = 0; i < GX; ++i)
        __builtin_nontemporal_store(u4{v[i].d[0], v[i].d[1], v[i].d[2], v[i].d[3]}, reinterpret_cast<u4*>(d + 8 * (lane + 64 * i)));
    } else if constexpr (MODE == 1) {  // dpp
      u4 v[GX + 1];
#pragma unroll
      for (int i = 0; i < GX; ++i) v[i] = *reinterpret_cast<const u4*>(s + 8 * (lane + 64 * i));
      v[GX] = u4{0, 0, 0, 0};
      if (lane == 0) v[GX] = *reinterpret_cast<const u4*>(s + 8 * (64 * GX));  // the group after the wave's last
      const unsigned q = sh >> 1, hb = (sh & 1) * 16;
#pragma unroll
      for (int i = 0; i < GX; ++i) {
        // dwords of the next group: lane + 1's, or lane 0's of the next 64-group stride for lane 63
        unsigned nx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned a = dpp<0x134>(v[i][k]);      // wave_rol:1: lane l reads lane l + 1 (63 reads 0)
          const unsigned b = dpp<0x134>(v[i + 1][k]);
          nx[k] = lane == 63 ? b : a;
        }
        const unsigned D[9] = {v[i][0], v[i][1], v[i][2], v[i][3], nx[0], nx[1], nx[2], nx[3], 0u};
        u4 o;
        switch (q) {  // wave-uniform
          case 0:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbit(D[k + 1], D[k], hb);
            break;
          case 1:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbit(D[k + 2], D[k + 1], hb);
            break;
          case 2:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbit(D[k + 3], D[k + 2], hb);
            break;
          default:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbit(D[k + 4], D[k + 3], hb);
            break;
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u4*>(d + 8 * (lane + 64 * i)));
      }
    } else if constexpr (MODE == 2) {  // two aligned loads per lane
      u4 a[GX], b[GX];
#pragma unroll
      for (int i = 0; i < GX; ++i) {
        a[i] = *reinterpret_cast<const u4*>(s + 8 * (lane + 64 * i));
        b[i] = *reinterpret_cast<const u4*>(s + 8 * (lane + 64 * i) + 8);
      }
      const unsigned q = sh >> 1, hb = (sh & 1) * 16;
#pragma unroll
      for (int i = 0; i < GX; ++i) {
        const unsigned D[9] = {a[i][0], a[i][1], a[i][2], a[i][3], b[i][0], b[i][1], b[i][2], b[i][3], 0u};
        unsigned E[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) E[k] = __builtin_amdgcn_alignbit(D[k + 1], D[k], hb);
        unsigned F[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) F[k] = (q & 1) ? E[k + 1] : E[k];
        u4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (q & 2) ? F[k + 2] : F[k];
        __builtin_nontemporal_store(o, reinterpret_cast<u4*>(d + 8 * (lane + 64 * i)));
      }
    } else {  // dword-aligned 16-byte load + one dword
      Q8 v[GX];
      unsigned e[GX];
      const int s4 = sh & ~1;
#pragma unroll
      for (int i = 0; i < GX; ++i) {
        v[i] = *reinterpret_cast<const Q8*>(s + s4 + 8 * (lane + 64 * i));
        e[i] = *reinterpret_cast<const unsigned*>(s + s4 + 8 * (lane + 64 * i) + 8);
      }
      const unsigned hb = (sh & 1) * 16;
#pragma unroll
      for (int i = 0; i < GX; ++i) {
        u4 o;
        o[0] = __builtin_amdgcn_alignbit(v[i].d[1], v[i].d[0], hb);
        o[1] = __builtin_amdgcn_alignbit(v[i].d[2], v[i].d[1], hb);
        o[2] = __builtin_amdgcn_alignbit(v[i].d[3], v[i].d[2], hb);
        o[3] = __builtin_amdgcn_alignbit(e[i], v[i].d[3], hb);
        __builtin_nontemporal_store(o, reinterpret_cast<u4*>(d + 8 * (lane + 64 * i)));
      }
    }
  }
}

int main(int argc, char** argv) {
  unsigned* dsem;
  CK(hipMalloc(&dsem, 256 * 4));
  k_dpp_semantics<<<1, 64>>>(dsem);
  unsigned hsem[256];
  CK(hipMemcpy(hsem, dsem, sizeof(hsem), hipMemcpyDeviceToHost));
  const char* names[4] = {"wave_shl:1 (0x130)", "wave_rol:1 (0x134)", "wave_shr:1 (0x138)", "wave_ror:1 (0x13c)"};
  for (int m = 0; m < 4; ++m)
    printf("%s: lane0<-%u lane1<-%u lane31<-%u lane32<-%u lane62<-%u lane63<-%u\n", names[m], hsem[64 * m], hsem[64 * m + 1],
           hsem[64 * m + 31], hsem[64 * m + 32], hsem[64 * m + 62], hsem[64 * m + 63]);

  const int n_rows = 16384, groups = 64 * GX * 16;  // rows of 32768 pixels = 64 KiB; 1 GiB each way
  const int64_t pitch = (int64_t)groups * 8 + 64;   // slack for the shifted reads
  const size_t bytes = (size_t)n_rows * pitch * 2 + 4096;
  uint16_t *src, *dst;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  std::vector<uint16_t> h((size_t)n_rows * pitch);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(i * 2654435761u >> 7);
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const double moved = 2.0 * n_rows * groups * 16;
  const char* mode_names[4] = {"unaligned 16 B loads", "aligned + DPP realign", "two aligned loads", "dword-aligned 16 B + 4 B"};
  std::vector<uint16_t> out((size_t)n_rows * pitch);
  for (int shift = 0; shift < 2; ++shift) {
    for (int mode = 0; mode < 4; ++mode) {
      auto launch = [&] {
        const dim3 grid(8192), block(256);
        switch (mode) {
          case 0: k_copy_rows<0><<<grid, block>>>(src, pitch, dst, pitch, n_rows, groups, shift); break;
          case 1: k_copy_rows<1><<<grid, block>>>(src, pitch, dst, pitch, n_rows, groups, shift); break;
          case 2: k_copy_rows<2><<<grid, block>>>(src, pitch, dst, pitch, n_rows, groups, shift); break;
          default: k_copy_rows<3><<<grid, block>>>(src, pitch, dst, pitch, n_rows, groups, shift); break;
        }
      };
      CK(hipMemset(dst, 0, bytes));
      launch();
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(out.data(), dst, out.size() * 2, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (int r = 0; r < n_rows; r += 97)
        for (int x = 0; x < groups * 8; ++x) bad += out[(size_t)r * pitch + x] != h[(size_t)r * pitch + x + ((shift + r) & 7)];
      hipEvent_t a, b;
      CK(hipEventCreate(&a));
      CK(hipEventCreate(&b));
      const int reps = 10;
      CK(hipEventRecord(a));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("shift0=%d %-26s %7.3f ms %7.1f GB/s  wrong pixels in sampled rows: %zu\n", shift, mode_names[mode], ms / reps,
             moved / (ms / reps * 1e-3) / 1e9, bad);
      fflush(stdout);
    }
  }
  return 0;
}
