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
// graph's N nodes, through an exchange buffer in global memory with one slab per stage.
//
// Round 3: no barrier, no flag, no cache maintenance -- DATA-TAGGED GRANULES (MI355X_MICROARCH.md, "handoff-1to1" /
// form R2 of "Workgroup dispatch ... visibility").  A row travels as F 8-byte words {value, tag}, written and read with
// relaxed agent-scope 8-byte atomics (global_store / global_load ... sc1: write-through, served past the reader's L1);
// tag = 16 * epoch + stage + 1 (at most 9 stages).  A consumer polls the words of its in-neighbours' rows THEMSELVES until every tag is the
// one it waits for: 8-byte accesses are single-copy atomic, so a word is either the old one (wrong tag) or the new one,
// and nothing has to be ordered against anything else.  Until round 2 each of the L + 1 hand-overs was a grid barrier
// with an agent-scope release (L2 write-back, ~1.7 us), an arrival atomic, a poll and an acquire (L1 invalidate, ~1.7 us):
// 30.7 us per predict at 20 links; the fence-free barrier with write-through rows (sc1 stores, acknowledgement wait, arrival
// atomic, poll, sc1 gathers) measured 25.4 us (and 18.5 against 13.7 us at 4 links: the acknowledgement of a write-through
// store is slower than a clean write-back); tagged granules: see profiles/HISTORY.md 3d.
// The epoch is floor(departures / N): every workgroup adds 1 to its graph's departure counter when it leaves (no return
// value, nobody waits for it), so the quotient is the same for all workgroups of a launch whenever they read it and one
// more in the next launch -- also under hipGraph replay, where the kernel arguments are frozen.  One slab per stage:
// a node that runs ahead never overwrites a row a slower reader still needs.
#pragma once
#include "kernels.hpp"
#include "kernels_fused.hpp"

namespace v2x {

struct SmallFwdArgs {
  const float* xe; const int32_t* row_ptr; const int32_t* col_idx;
  const float* params;
  int64_t gnn_off[FZ_MAXL + 1], gnn_sstride[FZ_MAXL + 1];      // flat-parameter offset of stage s (slot 0), slot stride
  int64_t dense_off[4], dense_sstride[4];
  unsigned long long* hbuf;                                    // [L + 1][SMALL rows][F] tagged words {value, tag}: the stage outputs
  unsigned long long* sync;                                    // [n_graphs]: departures of all launches so far (64-bit: never wraps)
  int* err;                                                    // contract-violation / exchange-timeout flag word (check_flag)
  int slab_rows;                                               // rows of one stage slab
  float* q;                                                    // [n_rows][C]
  int N, L, S, C, Dn, De, n_rows;
};

constexpr int SM_THREADS = 256;
constexpr int SM_BLOCK = SM_THREADS;
// Error bits of the flag word (host side: check_flag).  A poll that does not see its tags after SM_POLL_CAP rounds -- every
// round is at least one trip through memory, ~1 us: a quarter of a second, where a healthy hand-over takes ~2 us -- gives up,
// raises SM_ERR_TIMEOUT and lets the workgroup run to its end on whatever it read: the launch always terminates, the
// departure counter stays a multiple of N, the results are reported invalid (V2X_ESTATE) and the host re-arms the exchange.
constexpr int SM_ERR_SOURCE = 4 << 4;                    // a source id outside its graph (same bit as k_validate_batch)
constexpr int SM_ERR_TIMEOUT = 1 << 8;
constexpr int SM_POLL_CAP = 1 << 18;

// One layer's matrix-vector product, split so that the weights travel while the workgroup waits for something else (the
// grid barrier, the previous layer's reduction): `request` = all of this thread's rows (<= RMAX; clamped addresses
// beyond K, row K is the bias) into registers, `partial` = multiply with the input vector from LDS, `reduce` (after a
// workgroup barrier) = sum of the k-groups in fixed order.
// out[n] = sum_k v[k] * W[k][n] + W[K][n], n < NOUT; W row-major with NOUT columns.  NP = power of two >= NOUT lanes
// per k-group, 256 / NP k-groups take k = kg, kg + KG, ...; partial sums meet in `part` ([KG][NP]).
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
  __device__ __forceinline__ void partial(const float* v, int K, float* part) const {
    const int kg = threadIdx.x / NP;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int k = kg + r * KG;
      const float t = k < K ? v[k] : 0.f;
      if (r & 1) acc1 = fmaf(t, w[r], acc1); else acc0 = fmaf(t, w[r], acc0);
    }
    part[threadIdx.x] = acc0 + acc1;
  }
  __device__ __forceinline__ float reduce(const float* part) const {       // valid in threads < NOUT
    float out = bias;
    if (threadIdx.x < NP) {
#pragma unroll
      for (int g = 0; g < KG; ++g) out += part[g * NP + threadIdx.x];
    }
    return out;
  }
};

// tagged 8-byte words of the exchange buffer (relaxed agent-scope atomics lower to global_store / global_load ... sc1)
__device__ __forceinline__ void st_tagged(unsigned long long* p, float x, unsigned tag) {
  __hip_atomic_store(p, (unsigned long long)__float_as_uint(x) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_tagged(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int F>
__global__ __launch_bounds__(SM_BLOCK) void k_predict_small(SmallFwdArgs a) {
  constexpr int NPF = F;                                 // F is 16, 32 or 64: a power of two
  constexpr int KGF = SM_THREADS / NPF;
  __shared__ float sx[XE];                               // [x | e | pad]
  __shared__ __attribute__((aligned(16))) float sv[2 * F + XE];   // the current layer's input vector, real-row order
  __shared__ __attribute__((aligned(16))) float sh[F];            // this node's h_s
  __shared__ float sz[128];                              // Dense activations
  __shared__ float part[SM_THREADS];
  __shared__ int sNb[32];                                // the node's in-neighbours (N <= 32)
  __shared__ float sAgg[F];                              // the second half's partial neighbour sum
  const int q = blockIdx.x, g = blockIdx.y, N = a.N, L = a.L, C = a.C, xr = a.Dn + a.De;
  const int row = g * N + q, slot = a.S == 1 ? 0 : q, tid = threadIdx.x;
  const bool xw_wave = tid < 128;                        // waves 0 and 1 also run the exchange: in-neighbours 0..15 / 16..31
  const int xl = tid & 63, xh = tid >> 6;                // lane = feature, half
  unsigned long long* departures = a.sync + g;
  typedef const __attribute__((address_space(1))) float* gfp;

  // ---- requests that depend on nothing: the node's features, its CSR row, the embed weights
  // (the quotient of a 64-bit count: a 32-bit one would jump at its wrap in the middle of a launch whenever N is not a power
  //  of two -- after 2^32 / N predicts per graph slot; the tag keeps its low 28 bits, which wrap together for all workgroups)
  const unsigned epoch8 = 16u * (unsigned)((__hip_atomic_load(departures, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (unsigned long long)N) & 0x0fffffffull);
  const int e0 = a.row_ptr[row], deg = max(0, min(a.row_ptr[row + 1] - e0, 32));
  if (tid < XE) sx[tid] = a.xe[(int64_t)row * XE + tid];
  if (tid < deg) {
    // device-resident batches do not pass the host-side contract check: a source outside [0, N) would make this node poll a
    // row of ANOTHER graph (whose tags never become this graph's) -- clamp it and report it instead
    const int c = a.col_idx[e0 + tid];
    if ((c < 0 || c >= N) && a.err) atomicOr(a.err, SM_ERR_SOURCE);
    sNb[tid] = min(max(c, 0), N - 1);
  }
  Gemv<NPF, (XE + KGF - 1) / KGF> g0;
  // stage 0: rows 0..xr-1 = [x | e], rows xr..xr+F-1 = neighbour init (zero input in the reference: skipped), then bias
  const float* W0 = a.params + a.gnn_off[0] + slot * a.gnn_sstride[0];
  g0.request(W0, xr, F);
  g0.bias = ((gfp)W0)[(int64_t)(xr + F) * F + min(tid % NPF, F - 1)];
  Gemv<NPF, (2 * F + XE + KGF - 1) / KGF> gs;           // the graph layers, one after the other
  if (L >= 1) gs.request(a.params + a.gnn_off[1] + slot * a.gnn_sstride[1], 2 * F + xr, F);
  __syncthreads();
  g0.partial(sx, xr, part);
  __syncthreads();
  {
    const float o = g0.reduce(part);
    if (tid < F) sh[tid] = fmaxf(o, 0.f);
  }
  // the decision MLP's weights travel during the last barrier
  Gemv<128, (2 * F + XE + 1) / 2> d0;
  Gemv<64, H1 / 4> d1;
  Gemv<32, H2 / 8> d2;
  Gemv<16, (H3 + 15) / 16> d3;
  // ---- stages 1..L and the final aggregation
  for (int s = 1; s <= L + 1; ++s) {
    __syncthreads();                                     // sh = h_{s-1} of this node is complete
    const int xw = s <= L ? xr : a.Dn;
    const unsigned tag = epoch8 + (unsigned)s;           // stage s - 1, + 1: never 0 (the buffer starts zeroed)
    unsigned long long* slab = a.hbuf + (int64_t)(s - 1) * a.slab_rows * F;
    if (tid < F) st_tagged(slab + (int64_t)row * F + tid, sh[tid], tag);      // publish h_{s-1}: nobody waits for it here
    // the weight rows of whatever follows travel while the graph's other nodes get there
    if (s == L + 1) {
      d0.request(a.params + a.dense_off[0] + slot * a.dense_sstride[0], 2 * F + a.Dn, H1);
      d1.request(a.params + a.dense_off[1] + slot * a.dense_sstride[1], H1, H2);
      d2.request(a.params + a.dense_off[2] + slot * a.dense_sstride[2], H2, H3);
      d3.request(a.params + a.dense_off[3] + slot * a.dense_sstride[3], H3, C);
    }
    float acc = 0.f;
    if (xw_wave && xl < F && 16 * xh < deg) {            // neighbour sum, ascending sources inside each half
      // every word requested before the first one is looked at; entries past the in-degree re-read the last neighbour;
      // again until every tag is this stage's
      const unsigned long long* base = slab + (int64_t)g * N * F + xl;
      unsigned long long t[16];
      bool ok;
      int rounds = 0;
      do {
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = ld_tagged(base + (int64_t)sNb[min(16 * xh + e, deg - 1)] * F);
        ok = true;
#pragma unroll
        for (int e = 0; e < 16; ++e) ok = ok && (unsigned)(t[e] >> 32) == tag;
      } while (!ok && ++rounds < SM_POLL_CAP);
      if (!ok && xl == 0 && a.err) atomicOr(a.err, SM_ERR_TIMEOUT);      // bounded: see SM_POLL_CAP
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (16 * xh + e < deg) acc += __uint_as_float((unsigned)t[e]);
    }
    if (xw_wave && xh == 1 && xl < F) sAgg[xl] = acc;
    if (tid < F) sv[tid] = sh[tid];
    if (tid < xw) sv[F + tid] = sx[tid];
    __syncthreads();
    // (in-neighbours 0..15) + (16..31): k_agg's ascending order up to 16 in-neighbours, one more rounding beyond
    if (tid < F) sv[F + xw + tid] = acc + sAgg[tid];
    __syncthreads();                                     // sv = [h_{s-1} | x (| e) | agg_{s-1}] is complete
    if (s > L) break;
    gs.partial(sv, 2 * F + xr, part);
    __syncthreads();
    float o = gs.reduce(part);
    if (s < L) gs.request(a.params + a.gnn_off[s + 1] + slot * a.gnn_sstride[s + 1], 2 * F + xr, F);
    if (tid < F) sh[tid] = s < L ? fmaxf(o, 0.f) : o;
  }
  // ---- decision MLP of this node: z0 = [h_L | x | a_L] (BS_brain.py:175-179)
  {
    d0.partial(sv, 2 * F + a.Dn, part);
    __syncthreads();
    float o = d0.reduce(part);
    if (tid < H1) sz[tid] = fmaxf(o, 0.f);
    __syncthreads();
    d1.partial(sz, H1, part);
    __syncthreads();
    o = d1.reduce(part);
    __syncthreads();
    if (tid < H2) sv[tid] = fmaxf(o, 0.f);
    __syncthreads();
    d2.partial(sv, H2, part);
    __syncthreads();
    o = d2.reduce(part);
    if (tid < H3) sz[tid] = fmaxf(o, 0.f);
    __syncthreads();
    d3.partial(sz, H3, part);
    __syncthreads();
    o = d3.reduce(part);
    if (tid < C) a.q[(int64_t)row * C + tid] = o;
  }
  // ---- leave: one more departure (the next launch's epoch); nobody waits for the add
  if (tid == 0) __hip_atomic_fetch_add(departures, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace v2x
