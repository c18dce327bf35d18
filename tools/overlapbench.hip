// Does fp32 MFMA overlap with HBM streaming on MI355X, and how data-dependent is the MFMA rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// each wave: per iteration NL float4 loads (streaming) and NM MFMAs fed by the PREVIOUS iteration's data
template <int NL, int NM, bool DO_LOAD, bool DO_MFMA, bool RANDOM_INIT = false, bool ILV = false>
__global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ in, float* __restrict__ out, size_t n4, int iters) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 cur[NL], nxt[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) cur[l] = (DO_LOAD || RANDOM_INIT) ? in[(i + l * stride) & (n4 - 1)] : make_float4(1.f + threadIdx.x, 0.5f, 0.25f, 2.f);
  i += NL * stride;
  for (int it = 0; it < iters; ++it) {
    if (DO_LOAD) {
#pragma unroll
      for (int l = 0; l < NL; ++l) nxt[l] = in[(i + l * stride) & (n4 - 1)];
      i += NL * stride;
      if (!ILV) __builtin_amdgcn_sched_barrier(0);
    }
    if (DO_MFMA) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float4 v = cur[m % NL];
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x + m, v.y, acc[m & 3], 0, 0, 0);
      }
      if (ILV) {      // one load per NM/NL MFMAs instead of a burst of NL loads in front of the MFMAs
#pragma unroll
        for (int l = 0; l < NL; ++l) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[0][0] += cur[l].x + cur[l].y + cur[l].z + cur[l].w;
    }
    if (DO_LOAD) {
#pragma unroll
      for (int l = 0; l < NL; ++l) cur[l] = nxt[l];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// no in-wave prefetch: load tile -> wait -> MFMAs; overlap only across the waves sharing a SIMD
template <int NL, int NM>
__global__ __launch_bounds__(256) void k_mix_nopf(const float4* __restrict__ in, float* __restrict__ out, size_t n4, int iters) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    float4 cur[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) cur[l] = in[(i + l * stride) & (n4 - 1)];
    i += NL * stride;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const float4 v = cur[m % NL];
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x + m, v.y, acc[m & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
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
  const size_t MB = 1 << 20, sz = 256 * MB, n4 = sz / 16;
  float *in, *out;
  hipMalloc(&in, sz); hipMalloc(&out, 64 * MB);
  float* h = (float*)malloc(sz);
  for (size_t i = 0; i < sz / 4; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(in, h, sz, hipMemcpyHostToDevice);
  // node-update shaped: 9 float4 per lane and tile, 144 MFMAs per tile; steady state (32 tiles/wave) and the real
  // kernel's length (3 tiles/wave)
  for (int it2 : {32, 3})
    for (int wgs : {256, 512, 1024}) {
      double tl = time_us([&] { hipLaunchKernelGGL((k_mix<9, 144, true, false>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
      double tm = time_us([&] { hipLaunchKernelGGL((k_mix<9, 144, false, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
      double tb = time_us([&] { hipLaunchKernelGGL((k_mix<9, 144, true, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
      double tr = time_us([&] { hipLaunchKernelGGL((k_mix<9, 144, false, true, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
      double ti = time_us([&] { hipLaunchKernelGGL((k_mix<9, 144, true, true, false, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
      printf("   both, loads INTERLEAVED with the MFMAs (sched_group_barrier): %7.1f us\n", ti);
      printf("   mfma only, RANDOM register data: %7.1f us (%6.1f TF)\n", tr, (double)wgs * 4 * it2 * 144 * 2048 / tr / 1e6);
      const double bytes = (double)wgs * 256 * 9 * 16 * (it2 + 1), flop = (double)wgs * 4 * it2 * 144 * 2048;
      printf("node-shaped iters %2d wgs %4d: loads %7.1f us (%5.2f TB/s) | mfma %7.1f us (%6.1f TF) | both %7.1f us (%5.2f TB/s, %6.1f TF)\n",
             it2, wgs, tl, bytes / tl / 1e6, tm, flop / tm / 1e6, tb, bytes / tb / 1e6, flop / tb / 1e6);
    }
  // same total work (256 x 32 tile-iterations per wave slot), spread over 1, 2, 4, 8 waves per SIMD
  for (int wps : {1, 2, 4, 8}) {
    const int wgs = 256 * wps, it2 = 32 / wps;
    double tp = time_us([&] { hipLaunchKernelGGL((k_mix<9, 144, true, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
    double tn = time_us([&] { hipLaunchKernelGGL((k_mix_nopf<9, 144>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, it2); });
    printf("fixed work, %d waves/SIMD x %2d tiles: in-wave prefetch %7.1f us | no prefetch %7.1f us\n", wps, it2, tp, tn);
  }
  const int iters = 64;
  for (int wgs : {256, 512, 768}) {
    double tl = time_us([&] { hipLaunchKernelGGL((k_mix<7, 40, true, false>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, iters); });
    double tm = time_us([&] { hipLaunchKernelGGL((k_mix<7, 40, false, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, iters); });
    double tb = time_us([&] { hipLaunchKernelGGL((k_mix<7, 40, true, true>), dim3(wgs), dim3(256), 0, 0, (const float4*)in, out, n4, iters); });
    const double bytes = (double)wgs * 256 * 7 * 16 * (iters + 1), flop = (double)wgs * 4 * iters * 40 * 2048;
    printf("wgs %4d: loads only %7.1f us (%5.2f TB/s) | mfma only (const data) %7.1f us (%6.1f TF) | both (random data) %7.1f us (%5.2f TB/s, %6.1f TF)\n",
           wgs, tl, bytes / tl / 1e6, tm, flop / tm / 1e6, tb, bytes / tb / 1e6, flop / tb / 1e6);
  }
  return 0;
}
