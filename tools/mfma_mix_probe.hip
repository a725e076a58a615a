// Stand-alone measurement: v_mfma_f64_16x16x4_f64 interleaved with float64 vector work (rotor updates)
// as the subpixel kernels do; prints nanoseconds per loop iteration for a few mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// NM matrix instructions and NR complex rotor updates (4 mul + 2 add each) per iteration
template <int NM, int NR, bool SELECT>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, double a0, double b0) {
  d4 acc[NM];
#pragma unroll
  for (int i = 0; i < NM; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double wx[NR > 0 ? NR : 1], wy[NR > 0 ? NR : 1];
  const double sx = 0.999999, sy = 1e-3;
#pragma unroll
  for (int r = 0; r < (NR > 0 ? NR : 1); ++r) {
    wx[r] = 1.0 + r;
    wy[r] = 0.5 * r + threadIdx.x * 1e-9;
  }
  const bool part = (threadIdx.x >> 4) & 1;
  double a = a0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int r = NR > 0 ? i % NR : 0;
      const double b = NR > 0 ? (SELECT ? (part ? wy[r] : wx[r]) : wx[r]) : b0;
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const double nx = wx[r] * sx - wy[r] * sy, ny = wx[r] * sy + wy[r] * sx;
      wx[r] = nx;
      wy[r] = ny;
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

#define RUN(NM, NR, SEL)                                                                                          \
  {                                                                                                               \
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_mix<NM, NR, SEL>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); }); \
    printf("%2d mfma + %d rotors%s: %7.1f ns per iteration (%5.1f ns per mfma)\n", NM, NR, SEL ? " (selected operand)" : "", \
           ms * 1e6 / iters, ms * 1e6 / iters / NM);                                                               \
  }

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount;  // one wave per SIMD
  double* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
  const int iters = 20000;
  RUN(12, 0, false)
  RUN(12, 3, false)
  RUN(12, 3, true)
  RUN(3, 3, true)
  RUN(6, 3, true)
  RUN(10, 2, true)
  (void)hipFree(out);
  return 0;
}
