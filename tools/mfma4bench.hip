// v_mfma_f32_4x4x1_16b_f32 on MI355X: operand layout, raw rate, and the rate of the node update of a 4-graph tile with
// per-slot weights streamed from L2 (one 1 KiB wave load per k step feeds 4 x GQ MFMAs) -- the small-share fused kernels'
// inner loop (kernels_fused4.hpp), measured before they were written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gvec_p;

__global__ void k_layout(float* out) {
  const int lane = threadIdx.x;
  // A = 100 * lane, B = lane: D tells which (A lane, B lane) pairs meet in which (lane, register)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 * lane), (float)(lane + 1), acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[lane * 4 + v] = acc[v];
}

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// One stage of the 4-graph tile: wave w owns 4 slots; per k step one float4 per lane of weights (fragment-major: 1 KiB per
// wave), one LDS float per lane (the activation; read as b128 per 4 k), 4 x GQ MFMAs.  RING float4 loads in flight.
template <int NW, int GQ, int RING>
__global__ __launch_bounds__(64 * NW) void k_stream(const float* w, float* out, int ksteps, int stages, int n_items) {
  __shared__ f32x4 sB[NW * 16 * 36 * GQ];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NW * 16 * 36 * GQ; i += 64 * NW) sB[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 acc[GQ][4];
#pragma unroll
  for (int g = 0; g < GQ; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < stages; ++s) {
    const int item = (s * NW + wv) % n_items;
    gvec_p wp = (gvec_p)w + (int64_t)item * ksteps * 64 + lane;
    f32x4 ring[RING];
#pragma unroll
    for (int r = 0; r < RING; ++r) ring[r] = wp[r * 64];
    for (int k0 = 0; k0 < ksteps; k0 += RING) {
      f32x4 bv[GQ][RING / 4];
#pragma unroll
      for (int g = 0; g < GQ; ++g)
#pragma unroll
        for (int q = 0; q < RING / 4; ++q) bv[g][q] = sB[((wv * GQ + g) * 16 + (lane >> 2)) * 36 + ((k0 >> 2) + q) % 36];
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        const f32x4 a = ring[r];
        const int kn = k0 + r + RING;
        ring[r] = wp[(kn < ksteps ? kn : r) * 64];
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
          const float b = bv[g][r / 4][r % 4];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[g][i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b, acc[g][i], 0, 0, 0);
        }
      }
    }
  }
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < GQ; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[g][i];
  out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
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

template <int NW, int GQ, int RING>
void run_stream(const float* w, float* out, int grid, int n_items) {
  const int ksteps = 144, stages = 8;
  double t = time_us([&] { hipLaunchKernelGGL((k_stream<NW, GQ, RING>), dim3(grid), dim3(64 * NW), 0, 0, w, out, ksteps, stages, n_items); });
  double t1 = time_us([&] { hipLaunchKernelGGL((k_stream<NW, GQ, RING>), dim3(grid), dim3(64 * NW), 0, 0, w, out, ksteps, 1, n_items); });
  const double per_stage = (t - t1) / (stages - 1);
  const double bytes = (double)grid * NW * ksteps * 1024, flops = (double)grid * NW * ksteps * 4 * GQ * 512;
  printf("  NW %2d GQ %d RING %2d grid %4d: %.2f us / stage  (launch with 1 stage %.1f us)  L2 %.1f TB/s  %.1f TF\n", NW, GQ, RING, grid, per_stage, t1,
         bytes / per_stage / 1e6, flops / per_stage / 1e6);
}

int main() {
  float* out; hipMalloc(&out, 64 << 20);
  {
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
    std::vector<float> h(256);
    hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
    printf("layout (A = 100 lane, B = lane + 1): D[lane][v] / 100 -> (A lane, B lane)\n");
    for (int lane : {0, 1, 2, 3, 4, 5, 17, 63}) {
      printf("  lane %2d:", lane);
      for (int v = 0; v < 4; ++v) {
        const float d = h[lane * 4 + v];
        int found = 0;
        for (int la = 0; la < 64 && !found; ++la)
          for (int lb = 0; lb < 64 && !found; ++lb)
            if (d == (float)(100 * la) * (float)(lb + 1) && !(la == 0)) { printf("  v%d = A[%d] x B[%d]", v, la, lb); found = 1; }
        if (!found) printf("  v%d = %g", v, d);
      }
      printf("\n");
    }
  }
  const int iters = 512;
  for (int wpc : {1, 2}) {
    const int grid = 256 * wpc;
    const double n = (double)grid * 4 * iters * 8;
    double t1 = time_us([&] { hipLaunchKernelGGL(k_rate<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    double t2 = time_us([&] { hipLaunchKernelGGL(k_rate<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    double t4 = time_us([&] { hipLaunchKernelGGL(k_rate<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    double t8 = time_us([&] { hipLaunchKernelGGL(k_rate<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("4x4x1_16b wg/CU %d: nacc=1 %.1f TF  nacc=2 %.1f  nacc=4 %.1f  nacc=8 %.1f TF\n", wpc, n * 512 / t1 / 1e6, n * 2 * 512 / t2 / 1e6,
           n * 4 * 512 / t4 / 1e6, n * 8 * 512 / t8 / 1e6);
  }
  // weights: 20 slot groups x 144 k x 1 KiB ~ 2.9 MB per stage image; 5 items per workgroup-stage
  const int n_items = 60;
  float* w; hipMalloc(&w, (size_t)n_items * 144 * 1024);
  hipMemset(w, 0, (size_t)n_items * 144 * 1024);
  for (int grid : {128, 256, 512}) {
    printf("grid %d\n", grid);
    run_stream<5, 1, 8>(w, out, grid, n_items);
    run_stream<5, 1, 16>(w, out, grid, n_items);
    run_stream<5, 2, 8>(w, out, grid, n_items);
    run_stream<5, 2, 16>(w, out, grid, n_items);
    run_stream<10, 1, 8>(w, out, grid, n_items);
    run_stream<10, 1, 16>(w, out, grid, n_items);
  }
  return 0;
}
