// Micro-benchmark: achievable read / copy bandwidth on MI355X for the access shapes used by the engine.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// read only: sum to defeat DCE, one float per thread written
template <int UNROLL>
__global__ void k_read(const float4* __restrict__ in, float* __restrict__ out, size_t n4) {
  float s = 0.f;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) { float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[0] = s;
}
// fragment-shaped: lane (kg = l>>4, j = l&15) reads 16 B at row (base + j*rstride), column (kb*16 + 4*kg) floats;
// a wave covers 16 rows x 64 B per instruction, KB instructions per row tile (row width = KB*16 floats)
template <int KB>
__global__ void k_frag(const float* __restrict__ in, float* __restrict__ out, int n_tiles, int rstride_rows, int width) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kg = lane >> 4;
  float s = 0.f;
  for (int t = blockIdx.x * 4 + wv; t < n_tiles; t += gridDim.x * 4) {
    // tile t: 16 rows: row = (t / rstride_rows)*16*rstride_rows + (t % rstride_rows) + j*rstride_rows  (per-node style)
    const size_t row = (size_t)(t / rstride_rows) * 16 * rstride_rows + (t % rstride_rows) + (size_t)j * rstride_rows;
    float4 v[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) v[kb] = *reinterpret_cast<const float4*>(in + row * width + kb * 16 + 4 * kg);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) s += v[kb].x + v[kb].y + v[kb].z + v[kb].w;
  }
  if (s == 123.456f) out[0] = s;
}
// row-shaped: 16 lanes read one full 256-B row (KB=4 blocks) -> wave instruction = 4 rows x 256 B
__global__ void k_rows(const float* __restrict__ in, float* __restrict__ out, int n_rows, int rstride_rows) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, rr = lane >> 4;
  float s = 0.f;
  for (int t = blockIdx.x * 4 + wv; t < n_rows / 16; t += gridDim.x * 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jrow = u * 4 + rr;
      const size_t row = (size_t)(t / rstride_rows) * 16 * rstride_rows + (t % rstride_rows) + (size_t)jrow * rstride_rows;
      v[u] = *reinterpret_cast<const float4*>(in + row * 64 + 4 * li);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (s == 123.456f) out[0] = s;
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

int main() {
  const size_t MB = 1 << 20;
  float *in, *out;
  hipMalloc(&in, 1024 * MB); hipMalloc(&out, 1024 * MB);
  hipMemset(in, 0, 1024 * MB); hipMemset(out, 0, 1024 * MB);
  for (size_t sz : {21 * MB, 48 * MB, 96 * MB, 512 * MB}) {
    size_t n4 = sz / 16;
    for (int grid : {256, 1024, 4096}) {
      double t = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4); });
      double tr = time_us([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, (const float4*)in, out, n4); });
      double tr8 = time_us([&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, (const float4*)in, out, n4); });
      printf("size %4zu MB grid %5d : copy %7.1f us (%5.2f TB/s r+w)   read4 %7.1f us (%5.2f TB/s)   read8 %7.1f us (%5.2f TB/s)\n", sz / MB, grid, t,
             2.0 * sz / t / 1e6, tr, sz / tr / 1e6, tr8, sz / tr8 / 1e6);
    }
  }
  // engine shapes: R = 81920 rows
  const int R = 81920;
  for (int rs : {1, 20}) {
    for (int grid : {256, 640, 1280}) {
      double t9 = time_us([&] { hipLaunchKernelGGL(k_frag<9>, dim3(grid), dim3(256), 0, 0, in, out, R / 16, rs, 144); });
      double t4 = time_us([&] { hipLaunchKernelGGL(k_frag<4>, dim3(grid), dim3(256), 0, 0, in, out, R / 16, rs, 64); });
      double tr = time_us([&] { hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, in, out, R, rs); });
      printf("row_stride %2d grid %4d : frag K=144 (47 MB) %6.1f us (%5.2f TB/s)  frag K=64 (21 MB) %6.1f us (%5.2f TB/s)  rows K=64 %6.1f us (%5.2f TB/s)\n",
             rs, grid, t9, R * 576.0 / t9 / 1e6, t4, R * 256.0 / t4 / 1e6, tr, R * 256.0 / tr / 1e6);
    }
  }
  // launch overhead: empty-ish kernel
  double te = time_us([&] { hipLaunchKernelGGL(k_read<4>, dim3(256), dim3(256), 0, 0, (const float4*)in, out, (size_t)0); }, 200);
  printf("empty kernel back-to-back: %.2f us\n", te);
  return 0;
}
