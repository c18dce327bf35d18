// Can per-slot weight images be streamed from L2 fast enough to feed fp32 MFMA in a GRAPH-MAJOR fused GNN layer?
//   Geometry of the real thing: 256 workgroups (one per CU, 4 waves = 1 per SIMD), each wave
// walks `slots_per_wave` slots per layer; per slot it needs the slot's [144][64] weight image as MFMA A fragments
// (36 float4 per lane, 36.8 KB per wave, "fragment-major" so every load is a coalesced 1 KiB wave access) for
// 144 x RT MFMAs (RT = 16-row tiles per weight fetch).  All workgroups read the SAME 20 x 36.8 KB per layer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gvec_p;

template <int RT, bool DO_LOAD, bool DO_MFMA, bool ILV = false>
__global__ __launch_bounds__(256, 1) void k_stream(const float* __restrict__ wpk, float* __restrict__ out, int n_slots,
                                                   int layers, int spw) {
  extern __shared__ float smem[];   // forces one workgroup per CU
  const long long c0 = clock64(), r0 = wall_clock64();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc[RT][4];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 b[RT][9];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int k = 0; k < 9; ++k) b[r][k] = (f32x4){1.f + lane + r, 0.5f + k, 0.25f, 2.f};
  f32x4 w0[36], w1[36];
  auto load = [&](f32x4 (&w)[36], int it) {
    const int layer = (it / spw) % layers, slot = (wv + 4 * (it % spw)) % n_slots;
    gvec_p p = (gvec_p)(wpk + ((size_t)(layer * n_slots + slot) * 36 * 64 + lane) * 4);
#pragma unroll
    for (int c = 0; c < 36; ++c) w[c] = DO_LOAD ? p[c * 64] : (f32x4){0.1f * c, 0.2f, 0.3f, 0.4f};
  };
  auto mfma = [&](const f32x4 (&w)[36]) {
    if (DO_MFMA) {
#pragma unroll
      for (int kb = 0; kb < 9; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < RT; ++r)
              acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kb * 4 + n][s], b[r][kb][s], acc[r][n], 0, 0, 0);
    } else {
#pragma unroll
      for (int c = 0; c < 36; ++c) acc[0][c & 3] += w[c];
    }
  };
  const int n_it = layers * spw;
  load(w0, 0);
  int it = 0;
#pragma unroll 1
  for (; it + 2 <= n_it; it += 2) {
    if (ILV) {
      // one global load per 4*RT MFMAs: a wave issues in order, and a burst of 36 x 1 KiB loads blocks it at the
      // texture-address unit (64 B/clk per CU, shared by the 4 waves) for ~2000 cycles before its first MFMA
      load(w1, it + 1);
      mfma(w0);
#pragma unroll
      for (int c = 0; c < 36; ++c) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0); }
      __builtin_amdgcn_sched_barrier(0);
      load(w0, it + 2 < n_it ? it + 2 : 0);
      mfma(w1);
#pragma unroll
      for (int c = 0; c < 36; ++c) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0); }
      __builtin_amdgcn_sched_barrier(0);
    } else {
    load(w1, it + 1); __builtin_amdgcn_sched_barrier(0);
    mfma(w0); __builtin_amdgcn_sched_barrier(0);
    load(w0, it + 2 < n_it ? it + 2 : 0); __builtin_amdgcn_sched_barrier(0);
    mfma(w1); __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (it < n_it) mfma(w0);
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < 4; ++n) s += acc[r][n];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + smem[threadIdx.x & 7];
  if (blockIdx.x == 17 && threadIdx.x == 0) {      // shader-clock cycles and 100 MHz ticks this wave lived
    long long* t = reinterpret_cast<long long*>(out + (1 << 19));
    t[0] = clock64() - c0; t[1] = wall_clock64() - r0;
  }
}

template <typename F>
double time_us(F f, int iters = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / iters;
}

double mhz(float* out) {
  long long t[2]; hipMemcpy(t, out + (1 << 19), 16, hipMemcpyDeviceToHost);
  return t[1] ? t[0] * 100.0 / t[1] : 0.0;
}

template <int RT>
void run(const float* w, float* out, int wgs, int layers, int spw) {
  const size_t lds = 100 * 1024;
  hipFuncSetAttribute((const void*)k_stream<RT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_stream<RT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_stream<RT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  double tl = time_us([&] { hipLaunchKernelGGL((k_stream<RT, true, false>), dim3(wgs), dim3(256), lds, 0, w, out, 20, layers, spw); });
  const double fl = mhz(out);
  double tm = time_us([&] { hipLaunchKernelGGL((k_stream<RT, false, true>), dim3(wgs), dim3(256), lds, 0, w, out, 20, layers, spw); });
  const double fm = mhz(out);
  double tb = time_us([&] { hipLaunchKernelGGL((k_stream<RT, true, true>), dim3(wgs), dim3(256), lds, 0, w, out, 20, layers, spw); });
  const double fb = mhz(out);
  hipFuncSetAttribute((const void*)k_stream<RT, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  double ti = time_us([&] { hipLaunchKernelGGL((k_stream<RT, true, true, true>), dim3(wgs), dim3(256), lds, 0, w, out, 20, layers, spw); });
  const double fi = mhz(out);
  printf("  shader clock MHz: loads %.0f | mfma %.0f | both %.0f | interleaved %.0f\n", fl, fm, fb, fi);
  const double bytes = (double)wgs * 4 * layers * spw * 36 * 1024, flop = (double)wgs * 4 * layers * spw * 144.0 * RT * 2048;
  printf("RT %d wgs %4d layers %d slots/wave %2d: loads %7.1f us (%5.1f TB/s from L2) | mfma %7.1f us (%6.1f TF) | both %7.1f us (%5.1f TB/s, %6.1f TF) | both, loads interleaved with MFMAs %7.1f us (%6.1f TF)\n",
         RT, wgs, layers, spw, tl, bytes / tl / 1e6, tm, flop / tm / 1e6, tb, bytes / tb / 1e6, flop / tb / 1e6, ti, flop / ti / 1e6);
}

int main() {
  const size_t n = (size_t)3 * 20 * 36 * 64 * 4;      // 3 layers x 20 slots of packed [144][64] images = 2.2 MB
  float *w, *out;
  hipMalloc(&w, n * 4); hipMalloc(&out, 1 << 22);
  float* h = (float*)malloc(n * 4);
  for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(w, h, n * 4, hipMemcpyHostToDevice);
  for (int spw : {5, 40}) {
    run<1>(w, out, 256, 2, spw);
    run<2>(w, out, 256, 2, spw);
  }
  run<1>(w, out, 256, 3, 5);
  run<1>(w, out, 512, 2, 5);
  return 0;
}
