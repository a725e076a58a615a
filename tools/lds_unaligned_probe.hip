// Does ds_read_b128 accept addresses that are only 4-byte aligned on gfx950 (compute queues run with the
// unaligned LDS access mode), and what does it cost next to four ds_read2_b32?  Lane l reads the 8 floats at
// dword offset off + 8 l (two 16-byte reads, the access pattern of a lane that owns 8 consecutive pixels and
// looks their flatfield values up in a row held in LDS).
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_unaligned_probe.hip -o tools/lds_unaligned_probe
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

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kN = 8192;  // floats in LDS

__device__ __forceinline__ f4 lds_read16(unsigned byte_addr) {
  f4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}

template <int MODE>  // 0: ds_read_b128 by inline asm, 1: what the compiler emits for a 4-byte aligned 16-byte read
__global__ __launch_bounds__(256) void k(const float* in, float* out, int off, int iters) {
  __shared__ float buf[kN];
  for (int i = threadIdx.x; i < kN; i += blockDim.x) buf[i] = in[i];
  __syncthreads();
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned base = (unsigned)(size_t)buf;  // LDS byte address
  for (int it = 0; it < iters; ++it) {
    const int e = (off + 8 * (int)threadIdx.x + 8 * it) & (kN - 1 - 15);  // keeps 16 floats below the end
    const int d = (e & ~7) + (off & 7);
    if constexpr (MODE == 0) {
      const f4 a = lds_read16(base + 4u * d), b = lds_read16(base + 4u * d + 16u);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] += a[i];
        acc[4 + i] += b[i];
      }
    } else {
      struct __attribute__((packed, aligned(4))) F8 { float v[8]; };
      const F8 a = *reinterpret_cast<const F8*>(buf + d);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += a.v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + i] = acc[i];
}

int main() {
  std::vector<float> h(kN);
  for (int i = 0; i < kN; ++i) h[i] = (float)(i % 1021);
  float *in, *out;
  const int blocks = 2048;
  CK(hipMalloc(&in, kN * 4));
  CK(hipMalloc(&out, (size_t)blocks * 256 * 8 * 4));
  CK(hipMemcpy(in, h.data(), kN * 4, hipMemcpyHostToDevice));
  std::vector<float> got(256 * 8), got2(256 * 8);
  for (int off = 0; off < 8; ++off) {
    k<0><<<1, 256>>>(in, out, off, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    k<1><<<1, 256>>>(in, out, off, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got2.data(), out, got2.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, bad2 = 0;
    for (int t = 0; t < 256; ++t)
      for (int i = 0; i < 8; ++i) {
        const int e = (off + 8 * t) & (kN - 1 - 15);
        const int d = (e & ~7) + (off & 7);
        bad += got[t * 8 + i] != h[d + i];
        bad2 += got2[t * 8 + i] != h[d + i];
      }
    printf("offset %d dwords: ds_read_b128 %s (%d wrong), compiler's reads %s (%d wrong)\n", off, bad ? "WRONG" : "ok", bad,
           bad2 ? "WRONG" : "ok", bad2);
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int iters = 4096;
  for (int mode = 0; mode < 2; ++mode)
    for (int off = 0; off < 8; off += (off == 0 ? 1 : 3)) {
      auto launch = [&] {
        if (mode == 0) k<0><<<blocks, 256>>>(in, out, off, iters);
        else k<1><<<blocks, 256>>>(in, out, off, iters);
      };
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      launch();
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      const double bytes = (double)blocks * 256 * iters * 32;
      printf("%s offset %d: %.3f ms, %.1f TB/s of LDS reads, %.1f B/clk/CU at 2.4 GHz\n", mode == 0 ? "ds_read_b128 x2" : "compiler      ", off, ms,
             bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
  return 0;
}
