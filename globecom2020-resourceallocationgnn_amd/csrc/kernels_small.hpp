// k_predict_small: the whole Q-network forward of a FEW graphs in one launch -- the rollout-path predict
// (BS.predict / predict_one_step, /root/reference/BS_brain.py:336,1108,1394: one graph per call).
//
// With per-node weights and one graph there is nothing to batch: every node row meets its own weight matrix once, so the
// work is 20 matrix-VECTOR products per layer (0.8 MFLOP per graph in all) and the MFMA kernels of the training path
// spend their time on 15 shadow rows per tile (k_gnn_fwd_fused: 38 us for one graph, the same as for sixteen, plus a
// second launch for the decision MLP).  Here a workgroup owns ONE node of ONE graph for the whole network:
//   embed -> [grid barrier -> neighbour sum from the exchange buffer -> node update] x L -> grid barrier -> neighbour sum
//   -> Dense 80-40-20-C,
// weights streamed once from L2 / HBM in their flat row-major layout (lane = output feature: coalesced rows), the
// K axis split over the waves and reduced through LDS in a fixed order.  The only cross-workgroup traffic is h_s of the
// graph's N nodes: a ping-pong exchange buffer in global memory and L + 1 grid barriers (agent-scope release / acquire on
// one arrival counter per graph; the N x B <= #CUs workgroups are co-resident by construction, the host checks it).
// The counters clean up after themselves: the last workgroup of a graph to leave the final barrier zeroes them.
#pragma once
#include "kernels.hpp"
#include "kernels_fused.hpp"

namespace v2x {

struct SmallFwdArgs {
  const float* xe; const int32_t* row_ptr; const int32_t* col_idx;
  const float* params;
  int64_t gnn_off[FZ_MAXL + 1], gnn_sstride[FZ_MAXL + 1];      // flat-parameter offset of stage s (slot 0), slot stride
  int64_t dense_off[4], dense_sstride[4];
  float* hbuf;                                                 // [2][n_rows][F] exchange of the stage outputs
  unsigned* sync;                                              // [n_graphs][2]: arrivals, departures (zero between launches)
  float* q;                                                    // [n_rows][C]
  int N, L, S, C, Dn, De, n_rows;
};

constexpr int SM_THREADS = 256;

// One layer's matrix-vector product, split in two so that the weights travel while the workgroup waits for something
// else (the grid barrier, the previous layer's reduction): `request` = all of this thread's rows (<= RMAX; clamped
// addresses beyond K, row K is the bias) into registers, `apply` = multiply with the input vector from LDS and reduce.
// out[n] = sum_k v[k] * W[k][n] + W[K][n], n < NOUT; W row-major with NOUT columns.  NP = power of two >= NOUT lanes
// per k-group, 256 / NP k-groups take k = kg, kg + KG, ...; partial sums meet in `part` ([KG][NP]), fixed order.
template <int NP, int RMAX>
struct Gemv {
  static constexpr int KG = SM_THREADS / NP;
  float w[RMAX], bias;
  __device__ __forceinline__ void request(const float* W, int K, int NOUT) {
    typedef const __attribute__((address_space(1))) float* gfp;
    gfp Wg = (gfp)W;
    const int n = min((int)(threadIdx.x % NP), NOUT - 1), kg = threadIdx.x / NP;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) w[r] = Wg[(int64_t)min(kg + r * KG, K) * NOUT + n];
    bias = Wg[(int64_t)K * NOUT + n];
  }
  __device__ __forceinline__ float apply(const float* v, int K, float* part) const {   // valid in threads < NOUT
    const int kg = threadIdx.x / NP;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int k = kg + r * KG;
      const float t = k < K ? v[k] : 0.f;
      if (r & 1) acc1 = fmaf(t, w[r], acc1); else acc0 = fmaf(t, w[r], acc0);
    }
    part[threadIdx.x] = acc0 + acc1;
    __syncthreads();
    float out = bias;
    if (kg == 0) {
#pragma unroll
      for (int g = 0; g < KG; ++g) out += part[g * NP + threadIdx.x];
    }
    __syncthreads();
    return out;
  }
};

// All workgroups of graph g have published their rows: agent-scope release (L2 write-back) / arrive / poll / acquire (L2
// invalidate).  Tried: exchange buffer and counters in hipDeviceMallocUncached memory with relaxed atomics and no
// cache maintenance -- 1.5 us faster per launch and WRONG (stale rows: plain loads from that allocation are still
// served by a cache on this stack), so the fences stay.
__device__ __forceinline__ void graph_barrier(unsigned* arrivals, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(arrivals, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the other threads' view: nothing cached from before
}

template <int F>
__global__ __launch_bounds__(SM_THREADS) void k_predict_small(SmallFwdArgs a) {
  constexpr int NPF = F;                                 // F is 16, 32 or 64: a power of two
  constexpr int KGF = SM_THREADS / NPF;
  __shared__ float sx[XE];                               // [x | e | pad]
  __shared__ float sv[2 * F + XE];                       // the current layer's input vector, real-row order
  __shared__ float sh[F];                                // this node's h_s
  __shared__ float sz[128];                              // Dense activations
  __shared__ float part[SM_THREADS];
  __shared__ int sNb[32];                                // the node's in-neighbours (N <= 32)
  const int q = blockIdx.x, g = blockIdx.y, N = a.N, L = a.L, C = a.C, xr = a.Dn + a.De;
  const int row = g * N + q, slot = a.S == 1 ? 0 : q, tid = threadIdx.x;
  unsigned* arrivals = a.sync + 2 * g;
  typedef const __attribute__((address_space(1))) float* gfp;
  typedef __attribute__((address_space(1))) float* gfw;

  // ---- requests that depend on nothing: the node's features, its CSR row, the embed weights
  const int e0 = a.row_ptr[row], deg = min(a.row_ptr[row + 1] - e0, 32);
  if (tid < XE) sx[tid] = a.xe[(int64_t)row * XE + tid];
  if (tid < deg) sNb[tid] = a.col_idx[e0 + tid];
  Gemv<NPF, (XE + KGF - 1) / KGF> g0;
  // stage 0: rows 0..xr-1 = [x | e], rows xr..xr+F-1 = neighbour init (zero input in the reference: skipped), then bias
  const float* W0 = a.params + a.gnn_off[0] + slot * a.gnn_sstride[0];
  g0.request(W0, xr, F);
  g0.bias = ((gfp)W0)[(int64_t)(xr + F) * F + min(tid % NPF, F - 1)];
  Gemv<NPF, (2 * F + XE + KGF - 1) / KGF> gs;           // the graph layers, one after the other
  if (L >= 1) gs.request(a.params + a.gnn_off[1] + slot * a.gnn_sstride[1], 2 * F + xr, F);
  __syncthreads();
  {
    float o = g0.apply(sx, xr, part);
    if (tid < F) {
      o = fmaxf(o, 0.f);
      sh[tid] = o;
      ((gfw)a.hbuf)[(int64_t)row * F + tid] = o;
    }
  }
  // the decision MLP's weights travel during the last barrier
  Gemv<128, (2 * F + XE + 1) / 2> d0;
  Gemv<64, H1 / 4> d1;
  Gemv<32, H2 / 8> d2;
  Gemv<16, (H3 + 15) / 16> d3;
  // ---- stages 1..L and the final aggregation
  for (int s = 1; s <= L + 1; ++s) {
    if (s == L + 1) {
      d0.request(a.params + a.dense_off[0] + slot * a.dense_sstride[0], 2 * F + a.Dn, H1);
      d1.request(a.params + a.dense_off[1] + slot * a.dense_sstride[1], H1, H2);
      d2.request(a.params + a.dense_off[2] + slot * a.dense_sstride[2], H2, H3);
      d3.request(a.params + a.dense_off[3] + slot * a.dense_sstride[3], H3, C);
    }
    graph_barrier(arrivals, (unsigned)(N * s));
    gfp hb = (gfp)(a.hbuf + (int64_t)((s - 1) & 1) * a.n_rows * F);        // h_{s-1} of every node
    const int xw = s <= L ? xr : a.Dn;
    if (tid < F) {                                       // neighbour sum, ascending sources (k_agg order)
      float acc = 0.f;
      for (int e = 0; e < deg; ++e) acc += hb[(int64_t)(g * N + sNb[e]) * F + tid];
      sv[F + xw + tid] = acc;
      sv[tid] = sh[tid];
    }
    if (tid < xw) sv[F + tid] = sx[tid];
    __syncthreads();
    if (s > L) break;
    float o = gs.apply(sv, 2 * F + xr, part);
    if (s < L) gs.request(a.params + a.gnn_off[s + 1] + slot * a.gnn_sstride[s + 1], 2 * F + xr, F);
    if (tid < F) {
      if (s < L) o = fmaxf(o, 0.f);
      sh[tid] = o;
      ((gfw)a.hbuf)[(int64_t)(s & 1) * a.n_rows * F + (int64_t)row * F + tid] = o;
    }
  }
  // ---- decision MLP of this node: z0 = [h_L | x | a_L] (BS_brain.py:175-179)
  {
    float o = d0.apply(sv, 2 * F + a.Dn, part);
    if (tid < H1) sz[tid] = fmaxf(o, 0.f);
    __syncthreads();
    o = d1.apply(sz, H1, part);
    if (tid < H2) sv[tid] = fmaxf(o, 0.f);
    __syncthreads();
    o = d2.apply(sv, H2, part);
    if (tid < H3) sz[tid] = fmaxf(o, 0.f);
    __syncthreads();
    o = d3.apply(sz, H3, part);
    if (tid < C) a.q[(int64_t)row * C + tid] = o;
  }
  // ---- leave: the last workgroup of the graph zeroes the counters for the next launch
  if (tid == 0) {
    // (every workgroup is past the last barrier when it gets here, so relaxed atomics do; the kernel boundary publishes the zeros)
    const unsigned old = __hip_atomic_fetch_add(arrivals + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)N - 1) {
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(arrivals + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace v2x
