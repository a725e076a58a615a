// Streaming-copy ceilings of this GPU: what a kernel that only moves bytes reaches,
// for a few launch shapes.  Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_probe.hip -o tools/hbm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// one block moves U * 256 * 16 contiguous bytes, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_tile(const u4* __restrict__ src, u4* __restrict__ dst, size_t n) {
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  u4 v[U];
#pragma unroll
  for (int i = 0; i < U; ++i) {
    const size_t j = base + (size_t)i * 256;
    if (j < n) v[i] = NT ? __builtin_nontemporal_load(src + j) : src[j];
  }
#pragma unroll
  for (int i = 0; i < U; ++i) {
    const size_t j = base + (size_t)i * 256;
    if (j < n) {
      if (NT) __builtin_nontemporal_store(v[i], dst + j);
      else dst[j] = v[i];
    }
  }
}

// persistent grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_stride(const u4* __restrict__ src, u4* __restrict__ dst, size_t n) {
  const size_t step = (size_t)gridDim.x * (256 * U);
  for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += step) {
    u4 v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t j = base + (size_t)i * 256;
      if (j < n) v[i] = NT ? __builtin_nontemporal_load(src + j) : src[j];
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t j = base + (size_t)i * 256;
      if (j < n) {
        if (NT) __builtin_nontemporal_store(v[i], dst + j);
        else dst[j] = v[i];
      }
    }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const u4* __restrict__ src, u4* __restrict__ dst, size_t n) {
  const size_t step = (size_t)gridDim.x * (256 * U);
  u4 acc = {0, 0, 0, 0};
  for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += step) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t j = base + (size_t)i * 256;
      if (j < n) acc ^= NT ? __builtin_nontemporal_load(src + j) : src[j];
    }
  }
  if (acc.x == 0x12345678u && acc.y == 1) dst[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_write(u4* __restrict__ dst, size_t n) {
  const size_t step = (size_t)gridDim.x * (256 * U);
  const u4 v = {1, 2, 3, 4};
  for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += step) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t j = base + (size_t)i * 256;
      if (j < n) dst[j] = v;
    }
  }
}

template <class F>
static void run(const char* name, size_t bytes_moved, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms / reps, bytes_moved / (ms / reps * 1e-3) / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 6) << 30;
  const size_t n = bytes / 16;
  u4 *src, *dst;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes));
  CK(hipMemset(dst, 2, bytes));
  run("hipMemcpyDtoD", 2 * bytes, [&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); });
#define TILE(U, NT) \
  run("copy tile U=" #U " nt=" #NT, 2 * bytes, [&] { k_copy_tile<U, NT><<<(unsigned)((n + 256 * U - 1) / (256 * U)), 256>>>(src, dst, n); })
  TILE(1, false);
  TILE(2, false);
  TILE(4, false);
  TILE(8, false);
  TILE(4, true);
  TILE(8, true);
#define STRIDE(U, NT, G) \
  run("copy stride U=" #U " nt=" #NT " grid=" #G, 2 * bytes, [&] { k_copy_stride<U, NT><<<G, 256>>>(src, dst, n); })
  STRIDE(4, false, 2048);
  STRIDE(4, false, 4096);
  STRIDE(4, false, 8192);
  STRIDE(2, false, 4096);
  STRIDE(8, false, 2048);
  STRIDE(4, true, 4096);
  run("read U=4 grid=4096", bytes, [&] { k_read<4, false><<<4096, 256>>>(src, dst, n); });
  run("read U=8 grid=2048", bytes, [&] { k_read<8, false><<<2048, 256>>>(src, dst, n); });
  run("read nt U=4 grid=4096", bytes, [&] { k_read<4, true><<<4096, 256>>>(src, dst, n); });
  run("write U=4 grid=4096", bytes, [&] { k_write<4><<<4096, 256>>>(dst, n); });
  return 0;
}
