// Stand-alone measurement: sustained rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on one GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (double)i;
  const double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  double* out;
  hipMalloc(&out, (size_t)cus * 16 * 256 * sizeof(double));
  const int iters = 20000;
  printf("%d CUs, clock %d MHz\n", cus, prop.clockRate / 1000);
  for (int bpc : {1, 2, 4}) {
    const int blocks = cus * bpc;
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
      const double n = (double)blocks * 4 * iters * 8;  // wave-level instructions
      printf("mfma_f64_16x16x4: %d waves/SIMD: %.2f ms, %.1f TFLOP/s, %.1f ns per instruction and SIMD\n", bpc, ms,
             n * 2048 / ms / 1e9, ms * 1e6 / (n / (cus * 4)));
    }
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-9); });
      const double n = (double)blocks * 4 * iters * 16;
      printf("v_fma_f64:        %d waves/SIMD: %.2f ms, %.1f TFLOP/s, %.2f ns per instruction and SIMD\n", bpc, ms,
             n * 128 / ms / 1e9, ms * 1e6 / (n / (cus * 4)));
    }
  }
  hipFree(out);
  return 0;
}
