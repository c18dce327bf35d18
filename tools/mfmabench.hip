// Raw v_mfma_f32_16x16x4_f32 / 32x32x2 throughput on MI355X: NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double time_us(F f, int iters = 10) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / iters;
}
int main() {
  float* out; hipMalloc(&out, 64 << 20);
  const int iters = 512;
  for (int wgs_per_cu : {1, 2}) {
    const int grid = 256 * wgs_per_cu;
    const double n16 = (double)grid * 4 * iters * 8;       // MFMAs per accumulator-slot
    double t4 = time_us([&] { hipLaunchKernelGGL(k_mfma16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    double t8 = time_us([&] { hipLaunchKernelGGL(k_mfma16<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    double t1 = time_us([&] { hipLaunchKernelGGL(k_mfma16<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    double s4 = time_us([&] { hipLaunchKernelGGL(k_mfma32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("wg/CU %d: 16x16x4 nacc=1 %.1f TF  nacc=4 %.1f TF (%.1f us)  nacc=8 %.1f TF | 32x32x2 nacc=4 %.1f TF\n", wgs_per_cu,
           n16 * 1 * 2048 / t1 / 1e6, n16 * 4 * 2048 / t4 / 1e6, t4, n16 * 8 * 2048 / t8 / 1e6, n16 * 4 * 4096 / s4 / 1e6);
  }
  return 0;
}
