// Deep prefetch without registers: does an LDS-DMA ring (global_load_lds_dwordx4, D tiles in flight per wave) let ONE
// wave per SIMD overlap an HBM stream with fp32 MFMAs?  Same work as overlapbench's "node-shaped" case: per 16-row tile
// 9 x 1 KiB of operands (one float4 per lane each) and 144 MFMAs.  Register double buffering (overlapbench): loads 43 us,
// MFMAs 69 us, both 115 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NL, int NM, int D>
__global__ __launch_bounds__(256, 1) void k_ring(const float4* __restrict__ in, float* __restrict__ out, size_t n4, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const long long c0 = clock64(), r0 = wall_clock64();
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ring = (unsigned)(size_t)(smem) + wv * (D * NL * 1024);          // this wave's ring: D slots of NL KiB
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int slot) {
#pragma unroll
    for (int l = 0; l < NL; ++l) glds16(in + ((i + l * stride) & (n4 - 1)), ring + (slot * NL + l) * 1024);
    i += NL * stride;
  };
#pragma unroll
  for (int d = 0; d < D - 1; ++d) issue(d);
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
    issue((slot + D - 1) % D);                                          // tile it + D - 1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NL) : "memory");  // tile `it` has landed
    const f32x4* t = reinterpret_cast<const f32x4*>(smem + (wv * D + slot) * NL * 256) + lane;
    f32x4 v[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = t[l * 64];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m % NL][0] + m, v[m % NL][1], acc[m & 3], 0, 0, 0);
    slot = (slot + 1) % D;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 17 && threadIdx.x == 0) { long long* t = reinterpret_cast<long long*>(out + (1 << 20)); t[0] = clock64() - c0; t[1] = wall_clock64() - r0; }
}

template <typename F>
double time_us(F f, int iters = 10) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / iters;
}

template <int D>
void run(const float* in, float* out, size_t n4, int wgs, int iters) {
  auto k = k_ring<9, 144, D>;
  const size_t lds = (size_t)4 * D * 9 * 1024;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  double t = time_us([&] { hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, (const float4*)in, out, n4, iters); });
  const double bytes = (double)wgs * 256 * 9 * 16 * (iters + D - 1), flop = (double)wgs * 4 * iters * 144 * 2048;
  long long tt[2]; (void)hipMemcpy(tt, out + (1 << 20), 16, hipMemcpyDeviceToHost);
  printf("LDS ring depth %d, wgs %4d, %2d tiles/wave: %7.1f us (%5.2f TB/s, %6.1f TF)  shader clock %.0f MHz\n", D, wgs, iters, t, bytes / t / 1e6, flop / t / 1e6, tt[1] ? tt[0] * 100.0 / tt[1] : 0.0);
}

int main() {
  const size_t MB = 1 << 20, sz = 256 * MB, n4 = sz / 16;
  float *in, *out;
  (void)hipMalloc(&in, sz); (void)hipMalloc(&out, 64 * MB);
  float* h = (float*)malloc(sz);
  for (size_t i = 0; i < sz / 4; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  (void)hipMemcpy(in, h, sz, hipMemcpyHostToDevice);
  for (int wgs : {256}) {
    run<2>(in, out, n4, wgs, 32);
    run<3>(in, out, n4, wgs, 32);
    run<4>(in, out, n4, wgs, 32);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
