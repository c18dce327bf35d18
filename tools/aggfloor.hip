// Floor for the aggregation kernels' data movement at the headline size (81,920 rows x 64 floats): how long do the
// same bytes take with NO gather at all?   hipcc --offload-arch=gfx950 -O3 tools/aggfloor.hip -o /tmp/aggfloor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_fwd(const float4* in, float4* out, size_t n4) {      // read F, write F
  size_t i = blockIdx.x * (size_t)1024 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = in[min(i + 256 * u, n4 - 1)];
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) out[i + 256 * u] = v[u];
}
__global__ __launch_bounds__(256) void k_bwd(const float4* a, const float4* b, const float4* c, float4* out, size_t n4) {  // read 3F, write F
  size_t i = blockIdx.x * (size_t)1024 + threadIdx.x;
  float4 v[4], w[4], x[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { size_t k = min(i + 256 * u, n4 - 1); v[u] = a[k]; w[u] = b[k]; x[u] = c[k]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) {
    float4 r; r.x = x[u].x > 0 ? v[u].x + w[u].x : 0; r.y = x[u].y > 0 ? v[u].y + w[u].y : 0;
    r.z = x[u].z > 0 ? v[u].z + w[u].z : 0; r.w = x[u].w > 0 ? v[u].w + w[u].w : 0; out[i + 256 * u] = r; }
}
int main(int argc, char** argv) {
  const size_t n4 = 81920ull * 16;
  const size_t pad = argc > 1 ? atol(argv[1]) : 0;          // bytes of padding between the buffers
  float4 *a, *b, *c, *o;
  char* base; hipMalloc(&base, 4 * (n4 * 16 + pad) + 4096);
  a = (float4*)base; b = (float4*)(base + (n4 * 16 + pad)); c = (float4*)(base + 2 * (n4 * 16 + pad)); o = (float4*)(base + 3 * (n4 * 16 + pad));
  hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16); hipMemset(c, 0, n4 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = (int)((n4 + 1023) / 1024);
  for (int mm = 0; mm < 6; ++mm) { const int mode = mm & 1;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 100; ++it) {
        if (mode == 0) hipLaunchKernelGGL(k_fwd, dim3(blocks), dim3(256), 0, 0, a, o, n4);
        else hipLaunchKernelGGL(k_bwd, dim3(blocks), dim3(256), 0, 0, a, b, c, o, n4);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (mode == 0 ? 2.0 : 4.0) * n4 * 16;
      printf("%s: %.2f us/launch, %.2f TB/s (%d blocks)\n", mode == 0 ? "fwd-shaped copy" : "bwd-shaped fma ", 10.0 * ms, bytes / (ms * 1e-3 / 100) / 1e12, blocks);
    }
  }
  return 0;
}
