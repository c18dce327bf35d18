// CDNA4 (gfx950) device kernels of the V2X GNN Q-network engine.
//
// What each kernel replaces in the reference (/root/reference/BS_brain.py):
//   k_agg          AggLayer.call  :69-76   out[q] = sum_{p in N(q)} h[p]  (CSR gather, LDS-staged
//                                           graph tiles) and its transpose for the backward pass
//   k_gemm_rows    GNNLayer.call  :44-51   act([h|x]W1 + eW2 + aggW3 + b)  (fp32 MFMA 16x16x4)
//                                           + the data-gradient of the same layer
//   k_mlp_fwd/bwd  Dense x4       :176-179 80-40-20-C decision MLP, register-chained MFMA;
//                                           bwd fuses tf.losses.huber_loss (:86-87)
//   k_wgrad        TF autodiff of the K.dot's (weight gradients), fp32 MFMA over node rows
//   k_reduce_adam  keras.optimizers.Adam (:212), Keras 2.2.4 epsilon placement
//
// MFMA operand convention used everywhere ("swapped" GEMM, wave = 64 lanes):
//   v_mfma_f32_16x16x4_f32:  D[i][j] += sum_k A[i][k] B[k][j];  lane l supplies A[l&15][l>>4],
//   B[l>>4][l&15] and holds D[4*(l>>4)+r][l&15], r=0..3.
//   We put the WEIGHTS in A (i = output feature) and the ACTIVATIONS in B (j = node row), so a
//   lane ends up with 4 consecutive output features of ONE node row: a 16-byte store, and --
//   because the 16 k-values of a block are relabelled k = 4*(l>>4)+s for MFMA step s -- exactly
//   the B operand of the next layer.  The decision MLP therefore chains through registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace v2x {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gfloat_p;   // explicit GLOBAL pointer (global_load, counted vmcnt)
#define V2X_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int XE = 16;          // packed [x|e|pad] width
constexpr int H1 = 80, H2 = 40, H3 = 20;   // Dense widths (BS_brain.py:176-178)
constexpr int H2P = 48, H3P = 32, CP = 16; // padded to MFMA tiles
constexpr int LD1 = H1 + 4, LD2 = H2P + 4, LD3 = H3P + 4, LD4 = CP + 4;  // LDS strides = 4 mod 8

// padded-row -> real-row map of a weight matrix image: rows [pad_at, pad_at+n_pad) are zero pad
struct RowPad { int pad_at, n_pad, k_real; };
__device__ __forceinline__ int real_row(const RowPad& p, int rp) {
  int rr = rp < p.pad_at ? rp : (rp < p.pad_at + p.n_pad ? -1 : rp - p.n_pad);
  return rr < p.k_real ? rr : -1;
}

// copy W[k_real][n_real] (global, row-major) into an LDS image [KP][LD] with zero padding.
// Compile-time shape => constant divisors, fully unrolled; every global load is unconditional from an
// always-valid address (masked in registers afterwards) so that all PASSES loads are in flight at once:
// a conditional load makes hipcc branch around each one and serialises the L2 round trips.
template <int KP, int NP>
struct WeightImageRegs { float4 v[(KP * (NP / 4) + 255) / 256]; };

// the two halves of the copy, so that a caller with several images can have ALL their loads in flight together
template <int KP, int NP>
__device__ __forceinline__ void weight_image_load(WeightImageRegs<KP, NP>& r, const float* Wg, RowPad pad, int n_real, int tid) {
  constexpr int C4 = NP / 4;
  constexpr int TOTAL = KP * C4;
  constexpr int PASSES = (TOTAL + 255) / 256;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int i = tid + 256 * p;
    const int rp = i / C4, c = (i - rp * C4) << 2;
    const int rr = real_row(pad, rp);
    const bool ok = i < TOTAL && rr >= 0 && c < n_real;
    const float4 t = *reinterpret_cast<const float4*>(Wg + (ok ? (int64_t)rr * n_real + c : 0));
    const float mk = ok ? 1.f : 0.f;
    r.v[p] = make_float4(t.x * mk, t.y * mk, t.z * mk, t.w * mk);
  }
}
template <int KP, int NP, int LD>
__device__ __forceinline__ void weight_image_store(float* sW, const WeightImageRegs<KP, NP>& r, int tid) {
  constexpr int C4 = NP / 4;
  constexpr int TOTAL = KP * C4;
  constexpr int PASSES = (TOTAL + 255) / 256;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int i = tid + 256 * p;
    const int rp = i / C4, c = (i - rp * C4) << 2;
    if (i < TOTAL) *reinterpret_cast<float4*>(sW + rp * LD + c) = r.v[p];
  }
}
template <int KP, int NP, int LD>
__device__ __forceinline__ void fill_weight_image(float* sW, const float* Wg, RowPad pad, int n_real) {
  WeightImageRegs<KP, NP> r;
  weight_image_load<KP, NP>(r, Wg, pad, n_real, threadIdx.x);
  weight_image_store<KP, NP, LD>(sW, r, threadIdx.x);
}
__device__ __forceinline__ void fill_bias(float* sB, int np, const float* bg, int n_real) {
  for (int i = threadIdx.x; i < np; i += blockDim.x) sB[i] = i < n_real ? bg[i] : 0.f;
}

// =====================================================================================
// k_agg : neighbour gather + segment sum over CSR, one LDS tile per graph
// =====================================================================================
struct AggArgs {
  const float* src; int src_stride;     // rows to aggregate (width F at column 0 of src)
  const float* add; int add_stride;     // optional: out += add
  const float* mask;                    // optional [R][F]: out = mask > 0 ? out : 0 (ReLU')
  float* out;                           // [R][F]
  const int32_t* graph_off;             // [B+1] or null (fixed n_nodes)
  const int32_t* row_ptr;               // [R+1]
  const int32_t* col_idx;               // [E]
  int g_base, g_end;                    // graphs [g_base, g_end) of the batch are processed by this launch
  int n_nodes, F, lpr_shift;            // lanes per row = F/4 = 1 << lpr_shift
  int gpw;                              // graphs per workgroup (power of two)
  int rows_cap, edges_cap, mask_words;  // LDS capacities
  int transpose;                        // 0: out[q] = sum_{p->q} src[p];  1: out[p] = sum_{p->q} src[q]
  int* err;                             // host-visible flag word: bit 0 set when a tile exceeds the LDS capacities
};

// A workgroup whose graphs hold more rows / edges than the LDS tile was sized for (the batch's max_nodes / max_edges
// understate it) must not stage them: it raises the flag and leaves its output rows untouched.
#define V2X_TILE_GUARD(a, nrows, nedges)                                              \
  if ((nrows) > (a).rows_cap || (nedges) > (a).edges_cap || (nrows) < 0 || (nedges) < 0) { \
    if (threadIdx.x == 0 && (a).err) atomicOr((a).err, 1);                            \
    return;                                                                           \
  }

template <bool TRANSPOSE>   // forward gather (false) / its transpose for the backward pass (true)
__global__ __launch_bounds__(256) void k_agg(AggArgs a_in) {
  AggArgs a = a_in;
  a.transpose = TRANSPOSE ? 1 : 0;                            // compile-time from here on
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem;                                         // [rows_cap][F]
  int* sRp = reinterpret_cast<int*>(sT + a.rows_cap * a.F); // [rows_cap+1]
  int* sG = sRp + a.rows_cap + 1;                           // [gpw+1] local first row of each graph
  int* sCol = sG + a.gpw + 1;                               // [edges_cap]
  unsigned* sM = reinterpret_cast<unsigned*>(sCol + a.edges_cap);  // [rows_cap][mask_words]

  const int tid = threadIdx.x;
  const int g0 = a.g_base + blockIdx.x * a.gpw;
  const int g1 = min(g0 + a.gpw, a.g_end);
  const int r_begin = a.graph_off ? a.graph_off[g0] : g0 * a.n_nodes;
  const int r_end = a.graph_off ? a.graph_off[g1] : g1 * a.n_nodes;
  const int nrows = r_end - r_begin;
  const int e_begin = a.row_ptr[r_begin];
  const int nedges = a.row_ptr[r_end] - e_begin;
  const int LPR = 1 << a.lpr_shift;
  V2X_TILE_GUARD(a, nrows, nedges)

  // ---- stage: the WG's graphs are contiguous rows / contiguous edges => coalesced reads
  for (int i = tid; i < (nrows << a.lpr_shift); i += 256) {
    const int r = i >> a.lpr_shift, c = (i & (LPR - 1)) << 2;
    *reinterpret_cast<float4*>(sT + r * a.F + c) =
        *reinterpret_cast<const float4*>(a.src + (int64_t)(r_begin + r) * a.src_stride + c);
  }
  for (int i = tid; i <= nrows; i += 256) sRp[i] = a.row_ptr[r_begin + i] - e_begin;
  for (int i = tid; i < nedges; i += 256) sCol[i] = a.col_idx[e_begin + i];
  for (int i = tid; i <= a.gpw; i += 256) {
    const int g = min(g0 + i, g1);
    sG[i] = (a.graph_off ? a.graph_off[g] : g * a.n_nodes) - r_begin;
  }
  if (a.transpose)
    for (int i = tid; i < nrows * a.mask_words; i += 256) sM[i] = 0u;
  __syncthreads();

  if (a.transpose) {
    // transposed adjacency as per-source bit masks (integer atomics: order-independent result).
    // 32 threads per destination row, one edge each per pass: a short dependent chain instead of a
    // per-row serial loop over the in-edges.
    for (int i = tid; i < nrows * 32; i += 256) {
      const int r = i >> 5, sl = i & 31;
      int gi = 0;
      while (gi + 1 < a.gpw && sG[gi + 1] <= r) ++gi;
      const int gb = sG[gi], ql = r - gb;
      for (int e = sRp[r] + sl; e < sRp[r + 1]; e += 32)
        atomicOr(&sM[(gb + sCol[e]) * a.mask_words + (ql >> 5)], 1u << (ql & 31));
    }
    __syncthreads();
  }

  // ---- compute: a "worker" = LPR lanes owning one destination row at a time (float4 per lane)
  const int worker = tid >> a.lpr_shift, li4 = (tid & (LPR - 1)) << 2;
  const int nworkers = 256 >> a.lpr_shift;
  const int wpg = nworkers / a.gpw;            // workers per graph (>= 1 by host choice)
  const int gi = worker / wpg, wr = worker - gi * wpg;
  if (g0 + gi >= g1) return;
  const int gb = sG[gi], ng = sG[gi + 1] - gb;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = wr; q < ng; q += wpg) {
    const int64_t row = r_begin + gb + q;
    // epilogue operands first: their HBM latency overlaps the LDS gather below
    float4 addv = zero4, mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.add) addv = *reinterpret_cast<const float4*>(a.add + row * a.add_stride + li4);     // wave-uniform branch
    if (a.mask) mk = *reinterpret_cast<const float4*>(a.mask + row * a.F + li4);
    float4 acc = zero4;
    if (!a.transpose) {
      const int e1 = sRp[gb + q + 1];
      int e = sRp[gb + q];
      for (; e + 4 <= e1; e += 4) {   // 4 independent LDS gathers in flight
        const int p0 = sCol[e], p1 = sCol[e + 1], p2 = sCol[e + 2], p3 = sCol[e + 3];
        const float4 v0 = *reinterpret_cast<const float4*>(sT + (gb + p0) * a.F + li4);
        const float4 v1 = *reinterpret_cast<const float4*>(sT + (gb + p1) * a.F + li4);
        const float4 v2 = *reinterpret_cast<const float4*>(sT + (gb + p2) * a.F + li4);
        const float4 v3 = *reinterpret_cast<const float4*>(sT + (gb + p3) * a.F + li4);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
      }
      for (; e < e1; ++e) {
        const float4 v = *reinterpret_cast<const float4*>(sT + (gb + sCol[e]) * a.F + li4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    } else {
      for (int wd = 0; wd < a.mask_words; ++wd) {
        unsigned mbits = sM[(gb + q) * a.mask_words + wd];
        const float* base = sT + (gb + (wd << 5)) * a.F + li4;
        while (mbits) {               // ascending q; up to 4 independent LDS gathers in flight
          const int b0 = __builtin_ctz(mbits); mbits &= mbits - 1;
          const int b1 = mbits ? __builtin_ctz(mbits) : -1; if (mbits) mbits &= mbits - 1;
          const int b2 = mbits ? __builtin_ctz(mbits) : -1; if (mbits) mbits &= mbits - 1;
          const int b3 = mbits ? __builtin_ctz(mbits) : -1; if (mbits) mbits &= mbits - 1;
          const float4 v0 = *reinterpret_cast<const float4*>(base + b0 * a.F);
          const float4 v1 = *reinterpret_cast<const float4*>(base + (b1 < 0 ? b0 : b1) * a.F);
          const float4 v2 = *reinterpret_cast<const float4*>(base + (b2 < 0 ? b0 : b2) * a.F);
          const float4 v3 = *reinterpret_cast<const float4*>(base + (b3 < 0 ? b0 : b3) * a.F);
          const float m1 = b1 < 0 ? 0.f : 1.f, m2 = b2 < 0 ? 0.f : 1.f, m3 = b3 < 0 ? 0.f : 1.f;
          acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
          acc.x += v1.x * m1; acc.y += v1.y * m1; acc.z += v1.z * m1; acc.w += v1.w * m1;
          acc.x += v2.x * m2; acc.y += v2.y * m2; acc.z += v2.z * m2; acc.w += v2.w * m2;
          acc.x += v3.x * m3; acc.y += v3.y * m3; acc.z += v3.z * m3; acc.w += v3.w * m3;
        }
      }
    }
    acc.x += addv.x; acc.y += addv.y; acc.z += addv.z; acc.w += addv.w;
    acc.x = mk.x > 0.f ? acc.x : 0.f; acc.y = mk.y > 0.f ? acc.y : 0.f;
    acc.z = mk.z > 0.f ? acc.z : 0.f; acc.w = mk.w > 0.f ? acc.w : 0.f;
    *reinterpret_cast<float4*>(a.out + row * a.F + li4) = acc;
  }
}

// Small-tile form of k_agg (the 20-link graphs of the headline configuration): the workgroup's whole tile -- <= 1024
// float4 of features, <= 255 rows, <= 1024 edges, <= 4 destination rows per worker -- is fetched with EVERY global
// load issued before the first wait (features, row pointers, edge sources and the epilogue operands of all the
// worker's rows), then written to LDS.  In k_agg the same data arrives through 5+ dependent HBM round trips (the
// staging loops wait for each load before the LDS store), which is most of that kernel's 17-23 us.
// HAS_ADD / HAS_MASK are compile-time: `ptr ? load : 0` makes hipcc branch around every load and wait for it.
template <bool TRANSPOSE, bool HAS_ADD, bool HAS_MASK>
__global__ __launch_bounds__(256) void k_agg_small(AggArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem;                                         // [rows_cap][F]
  int* sRp = reinterpret_cast<int*>(sT + a.rows_cap * a.F); // [rows_cap+1]
  int* sCol = sRp + a.rows_cap + 1;                         // [edges_cap]
  unsigned* sM = reinterpret_cast<unsigned*>(sCol + a.edges_cap);  // [rows_cap][mask_words]

  const int tid = threadIdx.x;
  const int g0 = a.g_base + blockIdx.x * a.gpw;
  const int g1 = min(g0 + a.gpw, a.g_end);
  const int r_begin = a.graph_off ? a.graph_off[g0] : g0 * a.n_nodes;
  const int r_end = a.graph_off ? a.graph_off[g1] : g1 * a.n_nodes;
  const int nrows = r_end - r_begin;
  const int LPR = 1 << a.lpr_shift;
  const int nf4 = nrows << a.lpr_shift;
  if (nrows > a.rows_cap || nrows < 0) {           // (edge count checked below, once it is known)
    if (threadIdx.x == 0 && a.err) atomicOr(a.err, 1);
    return;
  }

  // worker = LPR lanes owning up to 4 destination rows of one graph
  const int worker = tid >> a.lpr_shift, li4 = (tid & (LPR - 1)) << 2;
  const int nworkers = 256 >> a.lpr_shift;
  const int wpg = nworkers / a.gpw;
  const int gi = worker / wpg, wr = worker - gi * wpg;
  const bool have_graph = g0 + gi < g1;
  const int gcl = min(g0 + gi, g1 - 1);
  const int gb = (a.graph_off ? a.graph_off[gcl] : gcl * a.n_nodes) - r_begin;
  const int ng = (a.graph_off ? a.graph_off[gcl + 1] : (gcl + 1) * a.n_nodes) - r_begin - gb;

  // ---- issue every global load.  (The empty asm statements "use" the loaded registers: without them LLVM sinks
  //      each load into the conditional LDS store that consumes it and the loads serialise again.)
#define V2X_PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
  // (wave-uniform addresses -> scalar loads: their wait does not hold up the vector loads below)
  const int e_begin = a.row_ptr[__builtin_amdgcn_readfirstlane(r_begin)];
  const int e_end = a.row_ptr[__builtin_amdgcn_readfirstlane(r_end)];
  float4 fv[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int i = min(tid + 256 * p, nf4 - 1);
    const int r = i >> a.lpr_shift, c = (i & (LPR - 1)) << 2;
    fv[p] = *reinterpret_cast<const float4*>(a.src + (int64_t)(r_begin + r) * a.src_stride + c);
  }
  int rpv = a.row_ptr[r_begin + min(tid, nrows)];
  float4 addv[4], mkv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t row = r_begin + gb + min(wr + k * wpg, ng - 1);
    if (HAS_ADD) addv[k] = *reinterpret_cast<const float4*>(a.add + row * a.add_stride + li4);
    else addv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_MASK) mkv[k] = *reinterpret_cast<const float4*>(a.mask + row * a.F + li4);
    else mkv[k] = make_float4(1.f, 1.f, 1.f, 1.f);
  }
  const int nedges = e_end - e_begin;
  if (nedges > a.edges_cap || nedges < 0) {        // workgroup-uniform: nothing has touched LDS yet
    if (threadIdx.x == 0 && a.err) atomicOr(a.err, 1);
    return;
  }
  int cv[4];
  const int e_safe = nedges > 0 ? e_begin : 0;        // an edge-less tile reads entry 0 (col_idx holds >= 1 entry)
#pragma unroll
  for (int p = 0; p < 4; ++p) cv[p] = a.col_idx[e_safe + min(tid + 256 * p, max(nedges - 1, 0))];
#pragma unroll
  for (int p = 0; p < 4; ++p) { V2X_PIN4(fv[p]); asm volatile("" : "+v"(cv[p])); }
  asm volatile("" : "+v"(rpv));
  if (HAS_ADD || HAS_MASK) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (HAS_ADD) V2X_PIN4(addv[k]); if (HAS_MASK) V2X_PIN4(mkv[k]); }
  }
#undef V2X_PIN4
  // ---- LDS
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int i = tid + 256 * p;
    if (i < nf4) *reinterpret_cast<float4*>(sT + (i >> a.lpr_shift) * a.F + ((i & (LPR - 1)) << 2)) = fv[p];
  }
  if (tid <= nrows) sRp[tid] = rpv - e_begin;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    if (tid + 256 * p < nedges) sCol[tid + 256 * p] = cv[p];
  if (TRANSPOSE)
    for (int i = tid; i < nrows * a.mask_words; i += 256) sM[i] = 0u;
  __syncthreads();

  if (TRANSPOSE) {
    // transposed adjacency as per-source bit masks: every edge is one integer atomicOr (order-independent)
    for (int i = tid; i < nrows * 32; i += 256) {
      const int r = i >> 5, sl = i & 31;
      int gj = 0, gbj = 0;
      if (a.graph_off) { while (gj + 1 < g1 - g0 && a.graph_off[g0 + gj + 1] - r_begin <= r) ++gj; gbj = a.graph_off[g0 + gj] - r_begin; }
      else { gj = r / a.n_nodes; gbj = gj * a.n_nodes; }
      const int ql = r - gbj;
      for (int e = sRp[r] + sl; e < sRp[r + 1]; e += 32)
        atomicOr(&sM[(gbj + sCol[e]) * a.mask_words + (ql >> 5)], 1u << (ql & 31));
    }
    __syncthreads();
  }
  if (!have_graph) return;

#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int q = wr + k * wpg;
    if (q >= ng) break;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!TRANSPOSE) {
      const int e1 = sRp[gb + q + 1];
      int e = sRp[gb + q];
      for (; e + 4 <= e1; e += 4) {
        const int p0 = sCol[e], p1 = sCol[e + 1], p2 = sCol[e + 2], p3 = sCol[e + 3];
        const float4 v0 = *reinterpret_cast<const float4*>(sT + (gb + p0) * a.F + li4);
        const float4 v1 = *reinterpret_cast<const float4*>(sT + (gb + p1) * a.F + li4);
        const float4 v2 = *reinterpret_cast<const float4*>(sT + (gb + p2) * a.F + li4);
        const float4 v3 = *reinterpret_cast<const float4*>(sT + (gb + p3) * a.F + li4);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
      }
      for (; e < e1; ++e) {
        const float4 v = *reinterpret_cast<const float4*>(sT + (gb + sCol[e]) * a.F + li4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    } else {
      for (int wd = 0; wd < a.mask_words; ++wd) {
        unsigned mbits = sM[(gb + q) * a.mask_words + wd];
        const float* base = sT + (gb + (wd << 5)) * a.F + li4;
        while (mbits) {
          const int b0 = __builtin_ctz(mbits); mbits &= mbits - 1;
          const int b1 = mbits ? __builtin_ctz(mbits) : -1; if (mbits) mbits &= mbits - 1;
          const int b2 = mbits ? __builtin_ctz(mbits) : -1; if (mbits) mbits &= mbits - 1;
          const int b3 = mbits ? __builtin_ctz(mbits) : -1; if (mbits) mbits &= mbits - 1;
          const float4 v0 = *reinterpret_cast<const float4*>(base + b0 * a.F);
          const float4 v1 = *reinterpret_cast<const float4*>(base + (b1 < 0 ? b0 : b1) * a.F);
          const float4 v2 = *reinterpret_cast<const float4*>(base + (b2 < 0 ? b0 : b2) * a.F);
          const float4 v3 = *reinterpret_cast<const float4*>(base + (b3 < 0 ? b0 : b3) * a.F);
          const float m1 = b1 < 0 ? 0.f : 1.f, m2 = b2 < 0 ? 0.f : 1.f, m3 = b3 < 0 ? 0.f : 1.f;
          acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
          acc.x += v1.x * m1; acc.y += v1.y * m1; acc.z += v1.z * m1; acc.w += v1.w * m1;
          acc.x += v2.x * m2; acc.y += v2.y * m2; acc.z += v2.z * m2; acc.w += v2.w * m2;
          acc.x += v3.x * m3; acc.y += v3.y * m3; acc.z += v3.z * m3; acc.w += v3.w * m3;
        }
      }
    }
    acc.x += addv[k].x; acc.y += addv[k].y; acc.z += addv[k].z; acc.w += addv[k].w;
    acc.x = mkv[k].x > 0.f ? acc.x : 0.f; acc.y = mkv[k].y > 0.f ? acc.y : 0.f;
    acc.z = mkv[k].z > 0.f ? acc.z : 0.f; acc.w = mkv[k].w > 0.f ? acc.w : 0.f;
    *reinterpret_cast<float4*>(a.out + (int64_t)(r_begin + gb + q) * a.F + li4) = acc;
  }
}

// =====================================================================================
// k_gemm_rows : node update  out = act(sum_seg in_seg . W_seg + b)  and its data-gradient
// =====================================================================================
struct GemmArgs {
  const float* seg0; int seg0_stride;   // h_prev (fwd) or dpre (dgrad), width F
  const float* xe;                      // [R][16]
  const float* seg2; int seg2_stride;   // agg_prev / neighbour-init, width F
  const float* W; int64_t slot_stride;  // layer base in the flat parameter buffer
  RowPad pad;
  float* out; int out_stride;
  int relu;
  int n_idx, row_stride, base_mul;      // row(idx) = (idx_base + idx)*row_stride + slot*base_mul
  int idx_base;
};

// Persistent tile loop shared by the node-update and MLP kernels: the workgroup's 4 waves walk a contiguous range of 16-row
// tiles (wave-interleaved); LOAD(t, buf) issues the global loads of tile t, COMPUTE(t, buf) consumes them.  The
// steady state is branch-free (loads of tile t+4 are issued, THEN tile t is computed under a counted vmcnt;
// sched_barrier pins that order) and the last one or two tiles are peeled.
// V2X_TILE_PROLOGUE issues the first tile's loads and is placed BEFORE the weight staging, so that both sets of
// loads are in flight together; V2X_TILE_LOOP follows the barrier.
#define V2X_TILE_PROLOGUE(T0, T_END, BUF_A, LOAD)                                     \
  const int t_ = (T0);                                                                \
  const int nt_ = t_ < (T_END) ? ((T_END) - t_ + 3) >> 2 : 0;                         \
  if (nt_ > 0) LOAD(t_, BUF_A);                                                       \
  __builtin_amdgcn_sched_barrier(0);

#define V2X_TILE_LOOP(BUF_A, BUF_B, LOAD, COMPUTE)                                    \
  {                                                                                   \
    int k_ = 0;                                                                       \
    for (; k_ + 2 < nt_; k_ += 2) {                                                   \
      LOAD(t_ + 4 * (k_ + 1), BUF_B);                                                 \
      __builtin_amdgcn_sched_barrier(0);                                              \
      COMPUTE(t_ + 4 * k_, BUF_A);                                                    \
      __builtin_amdgcn_sched_barrier(0);                                              \
      LOAD(t_ + 4 * (k_ + 2), BUF_A);                                                 \
      __builtin_amdgcn_sched_barrier(0);                                              \
      COMPUTE(t_ + 4 * (k_ + 1), BUF_B);                                              \
      __builtin_amdgcn_sched_barrier(0);                                              \
    }                                                                                 \
    if (nt_ - k_ == 2) {                                                              \
      LOAD(t_ + 4 * (k_ + 1), BUF_B);                                                 \
      __builtin_amdgcn_sched_barrier(0);                                              \
      COMPUTE(t_ + 4 * k_, BUF_A);                                                    \
      COMPUTE(t_ + 4 * (k_ + 1), BUF_B);                                              \
    } else if (nt_ - k_ == 1) {                                                       \
      COMPUTE(t_ + 4 * k_, BUF_A);                                                    \
    }                                                                                 \
  }

// Persistent form: grid = (workgroups per slot, slots).  A workgroup stages its slot's weight image in LDS
// ONCE, then its 4 waves walk a contiguous range of 16-row tiles (wave-interleaved), each wave keeping the
// NEXT tile's activation fragments in flight (register double buffer) while the MFMAs of the current tile
// run; no barrier in the loop.  The host sizes the grid so that every SIMD gets the same number of tiles.
template <int F, bool HAS0, bool HAS2, bool DGRAD>
__global__ __launch_bounds__(256) void k_gemm_rows(GemmArgs a) {
  constexpr int FB = F / 16;
  constexpr int KB = DGRAD ? FB : ((HAS0 ? FB : 0) + 1 + (HAS2 ? FB : 0));
  constexpr int NT = DGRAD ? 2 * FB : FB;
  constexpr int KP = DGRAD ? (2 * F + XE) : KB * 16;
  constexpr int LDW = F + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;              // [KP][LDW]
  float* sB = smem + KP * LDW;   // [F]

  const int slot = blockIdx.y;
  const float* Wg = a.W + slot * a.slot_stride;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, kg = lane >> 4;

  const int n_tiles = (a.n_idx + 15) >> 4;
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int t_end = min((int)(blockIdx.x + 1) * per, n_tiles);

  // B operand (activations) of one 16-row tile: one float4 per 16-wide K block, straight HBM -> VGPR.
  // Rows past n_idx are clamped (valid address, result dropped at the store).
  auto load_tile = [&](int t, float4 (&bf)[KB], int64_t& row) {
    const int idx = min(t * 16 + j, a.n_idx - 1);
    row = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
    int kb = 0;
    if (HAS0 || DGRAD) {
#pragma unroll
      for (int b = 0; b < FB; ++b)
        bf[kb++] = *reinterpret_cast<const float4*>(a.seg0 + row * a.seg0_stride + b * 16 + 4 * kg);
    }
    if (!DGRAD) {
      bf[kb++] = *reinterpret_cast<const float4*>(a.xe + row * XE + 4 * kg);
      if (HAS2) {
#pragma unroll
        for (int b = 0; b < FB; ++b)
          bf[kb++] = *reinterpret_cast<const float4*>(a.seg2 + row * a.seg2_stride + b * 16 + 4 * kg);
      }
    }
  };

  auto compute_store = [&](int t, const float4 (&bf)[KB], int64_t row) {
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float w[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (!DGRAD) {
          // A[i = out feature nt*16+j][k = kb*16 + 4*kg + s]  = W[k][i]  (column of the image)
#pragma unroll
          for (int s = 0; s < 4; ++s) w[nt][s] = sW[(kb * 16 + 4 * kg + s) * LDW + nt * 16 + j];
        } else {
          // A[i = input feature][k = out feature]  = W[i][k]  (row of the image, skipping the xe rows)
          const int orow = nt < FB ? nt * 16 : F + XE + (nt - FB) * 16;
          const float4 tw = *reinterpret_cast<const float4*>(sW + (orow + j) * LDW + kb * 16 + 4 * kg);
          w[nt][0] = tw.x; w[nt][1] = tw.y; w[nt][2] = tw.z; w[nt][3] = tw.w;
        }
      }
      // k-step outer, tile inner: consecutive MFMAs hit different accumulators (40-cycle dependent latency)
      const float bv[4] = {bf[kb].x, bf[kb].y, bf[kb].z, bf[kb].w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = V2X_MFMA(w[nt][s], bv[s], acc[nt]);
    }
    // epilogue: lane holds out[row j][nt*16 + 4*kg .. +3]
    if (t * 16 + j < a.n_idx) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 v = acc[nt];
        if (!DGRAD) {
          const float4 b = *reinterpret_cast<const float4*>(sB + nt * 16 + 4 * kg);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          if (a.relu) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          }
        }
        *reinterpret_cast<float4*>(a.out + row * a.out_stride + nt * 16 + 4 * kg) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };

  struct Tile { float4 bf[KB]; int64_t row; };
  auto load_t = [&](int t, Tile& x) { load_tile(t, x.bf, x.row); };
  auto comp_t = [&](int t, const Tile& x) { compute_store(t, x.bf, x.row); };
  Tile tA, tB;
  V2X_TILE_PROLOGUE(blockIdx.x * per + wv, t_end, tA, load_t)
  fill_weight_image<KP, F, LDW>(sW, Wg, a.pad, F);
  if (!DGRAD) fill_bias(sB, F, Wg + (int64_t)a.pad.k_real * F, F);
  __syncthreads();
  V2X_TILE_LOOP(tA, tB, load_t, comp_t)
}

// =====================================================================================
// decision MLP  (Dense 80 relu, 40 relu, 20 relu, C linear)  -- register-chained MFMA
// =====================================================================================
struct MlpArgs {
  const float* h; const float* xe; const float* agg;   // z0 = [h | x | agg]  (rows of Dense-0 permuted)
  const float* W[4]; int64_t slot_stride[4];            // flat-parameter bases of the 4 Dense layers
  int C;
  float* z1; float* z2; float* z3; float* q;            // [R][80], [R][40], [R][20], [R][C]
  // backward only
  const float* y; float inv_denom;
  float* dq; float* dz1; float* dz2; float* dz3; float* gha;   // gha[R][2F] = [dh | dagg]
  float* rowloss;                                       // [R] sum_c huber(y - q)
  int n_idx, row_stride, base_mul, idx_base;
  // z2, z3, dz2, dz3, dq and rowloss are only ever touched slot by slot (this kernel writes them, k_wgrad reads
  // them), so they are stored SLOT-MAJOR: row(slot, idx) = slot * srow_stride + idx.  In graph-major order a slot's
  // rows lie N rows apart and a 16..160-byte row costs a whole 128-byte line per access (measured: 30 % more HBM
  // traffic than the algorithmic bytes in the weight-gradient launch).  Shared weights: one slot, identical layout.
  int64_t srow_stride;
  // k_mlp_train_wg between the fused graph-layer kernels (per-node weights, whole 16-graph groups): h, agg and gha are
  // FRAGMENT-major, [slot][group][n-tile][64 lanes] float4 -- the value a lane of those kernels holds for (slot, group,
  // tile), so that a wave's access is 1 KiB contiguous instead of 64 pieces of 16 bytes from 16 rows (the texture
  // path handles a wave's request line by line).  0: row-major, else the number of groups per slot.
  int frag_groups;
  // k_mlp_train_wg inside a DQN replay step (v2x_dqn_step): the targets are formed in the kernel -- y = q (THIS forward's
  // output, as Keras' fit sees the prediction it was handed, BS_brain.py:664-692, :728) except y[action[row]] = the replaced
  // entry -- and written where q would go.  tq non-null says so; `y` then points at tq: one float4 per row, {replaced entry,
  // action (bits), -, -} (k_dqn_tq), read by the SAME unconditional float4 load as a row of targets (a load under a branch
  // would put a vmcnt(0) at the join in front of the next tile's prefetch: measured + 1.7 K cycles per tile).  The separate
  // online MLP forward and the target kernel's pass over q drop out.
  const float* tq;
};

template <int F>    // F == 0: "tail" form without the Dense-0 image (wide features: Dense-0 runs in k_wide_gemm)
struct MlpLds {
  static constexpr int K1P = F > 0 ? 2 * F + XE : 0;
  static constexpr int W1 = 0;
  static constexpr int W2 = W1 + K1P * LD1;
  static constexpr int W3 = W2 + H1 * LD2;
  static constexpr int W4 = W3 + H2P * LD3;
  static constexpr int B1 = W4 + H3P * LD4;
  static constexpr int B2 = B1 + H1;
  static constexpr int B3 = B2 + H2P;
  static constexpr int B4 = B3 + H3P;
  static constexpr int TOTAL = B4 + CP;   // floats
};

template <int F>
__device__ __forceinline__ void mlp_fill_lds(float* smem, const MlpArgs& a, int slot, bool with_bias, int ti) {
  using L = MlpLds<F>;
  const int C = a.C;
  const float* w1 = a.W[0] + slot * a.slot_stride[0];
  const float* w2 = a.W[1] + slot * a.slot_stride[1];
  const float* w3 = a.W[2] + slot * a.slot_stride[2];
  const float* w4 = a.W[3] + slot * a.slot_stride[3];
  const int k1 = 2 * F + (2 * C + 1);
  // Dense-0 rows [h(F) | x(2C+1) | agg(F)]: the image keeps xe's 16 columns; the edge-feature
  // and pad rows are ZERO so the packed xe block can be used as-is (BS_brain.py:175 feeds
  // only the node features to the decision DNN).
  // all four images' loads are issued before the first LDS store (measured: the four fills one after the other, each a
  // dependent L2 round trip, took 9 us of every MLP launch)
  WeightImageRegs<(F > 0 ? L::K1P : 16), H1> r1;
  WeightImageRegs<H1, H2P> r2;
  WeightImageRegs<H2P, H3P> r3;
  WeightImageRegs<H3P, CP> r4;
  if constexpr (F > 0) weight_image_load<L::K1P, H1>(r1, w1, RowPad{F + 2 * C + 1, XE - (2 * C + 1), k1}, H1, ti);
  weight_image_load<H1, H2P>(r2, w2, RowPad{H1, 0, H1}, H2, ti);
  weight_image_load<H2P, H3P>(r3, w3, RowPad{H2, H2P - H2, H2}, H3, ti);
  weight_image_load<H3P, CP>(r4, w4, RowPad{H3, H3P - H3, H3}, C, ti);
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (with_bias) {                          // one bias value per thread and layer (widths <= 80 < 256 threads)
    if constexpr (F > 0) bv[0] = w1[(int64_t)k1 * H1 + min(ti, H1 - 1)];
    bv[1] = w2[H1 * H2 + min(ti, H2 - 1)];
    bv[2] = w3[H2 * H3 + min(ti, H3 - 1)];
    bv[3] = w4[H3 * C + min(ti, C - 1)];
  }
  if constexpr (F > 0) weight_image_store<L::K1P, H1, LD1>(smem + L::W1, r1, ti);
  weight_image_store<H1, H2P, LD2>(smem + L::W2, r2, ti);
  weight_image_store<H2P, H3P, LD3>(smem + L::W3, r3, ti);
  weight_image_store<H3P, CP, LD4>(smem + L::W4, r4, ti);
  if (with_bias) {
    if constexpr (F > 0) { if (ti < H1) smem[L::B1 + ti] = bv[0]; }
    if (ti < H2P) smem[L::B2 + ti] = ti < H2 ? bv[1] : 0.f;
    if (ti < H3P) smem[L::B3 + ti] = ti < H3 ? bv[2] : 0.f;
    if (ti < CP) smem[L::B4 + ti] = ti < C ? bv[3] : 0.f;
  }
}

// acc[nt] (+)= W^T-tile x act-block  for one K block (column reads of the weight image)
template <int RT, int NT>
__device__ __forceinline__ void mfma_cols(const float* sW, int ld, int kb, int j, int kg,
                                          const f32x4 (&bblk)[RT], f32x4 (&acc)[RT][NT]) {
  // consecutive MFMAs go to DIFFERENT accumulators (k-step outer, tile inner): a dependent 16x16x4 f32 MFMA
  // has 40 cycles of latency against a 32-cycle issue interval
  float w[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int s = 0; s < 4; ++s) w[nt][s] = sW[(kb * 16 + 4 * kg + s) * ld + nt * 16 + j];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = V2X_MFMA(w[nt][s], bblk[rt][s], acc[rt][nt]);
}

// acc (+)= W-tile(rows orow..orow+15) x grad-block  for one K block (row reads of the image)
template <int RT>
__device__ __forceinline__ void mfma_rows(const float* sW, int ld, int orow, int kb, int j, int kg,
                                          const f32x4 (&bblk)[RT], f32x4 (&acc)[RT]) {
  const float4 t = *reinterpret_cast<const float4*>(sW + (orow + j) * ld + kb * 16 + 4 * kg);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    acc[rt] = V2X_MFMA(t.x, bblk[rt][0], acc[rt]);
    acc[rt] = V2X_MFMA(t.y, bblk[rt][1], acc[rt]);
    acc[rt] = V2X_MFMA(t.z, bblk[rt][2], acc[rt]);
    acc[rt] = V2X_MFMA(t.w, bblk[rt][3], acc[rt]);
  }
}

// acc[nt] (+)= W-tile(rows orow(nt)..+15) x grad-block for one K block, all NT output tiles of a layer at once
// (k-step outer, tile inner: independent accumulators back-to-back)
template <int NT, typename OROW>
__device__ __forceinline__ void mfma_rows_multi(const float* sW, int ld, OROW orow, int kb, int j, int kg,
                                                const f32x4& bblk, f32x4 (&acc)[NT][1]) {
  float4 t[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) t[nt] = *reinterpret_cast<const float4*>(sW + (orow(nt) + j) * ld + kb * 16 + 4 * kg);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt][0] = V2X_MFMA(t[nt].x, bblk[0], acc[nt][0]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt][0] = V2X_MFMA(t[nt].y, bblk[1], acc[nt][0]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt][0] = V2X_MFMA(t[nt].z, bblk[2], acc[nt][0]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt][0] = V2X_MFMA(t[nt].w, bblk[3], acc[nt][0]);
}

__device__ __forceinline__ f32x4 ld4(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  return (f32x4){t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
__device__ __forceinline__ f32x4 gate4(f32x4 g, f32x4 z) {   // g * (z > 0)
  return (f32x4){z[0] > 0.f ? g[0] : 0.f, z[1] > 0.f ? g[1] : 0.f, z[2] > 0.f ? g[2] : 0.f,
                 z[3] > 0.f ? g[3] : 0.f};
}

template <int F>
struct MlpFwdIn { f32x4 z0[F > 0 ? 2 * (F / 16) + 1 : 5]; int64_t row; };   // tail form: the 5 blocks of z1

// Persistent: grid = (workgroups per slot, slots); the slot's 4 weight images are staged in LDS once.
template <int F>
__global__ __launch_bounds__(256, 2) void k_mlp_fwd(MlpArgs a) {
  using L = MlpLds<F>;
  constexpr int FB = F / 16, KB1 = 2 * FB + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int slot = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int n_tiles = (a.n_idx + 15) >> 4;
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int t_end = min((int)(blockIdx.x + 1) * per, n_tiles);

  auto load_in = [&](int t, MlpFwdIn<F>& in) {
    const int idx = min(t * 16 + j, a.n_idx - 1);                 // clamped: always a valid row
    const int64_t row = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
    in.row = row;
    if constexpr (F > 0) {
#pragma unroll
      for (int b = 0; b < FB; ++b) in.z0[b] = ld4(a.h + row * F + b * 16 + 4 * kg);
      in.z0[FB] = ld4(a.xe + row * XE + 4 * kg);
#pragma unroll
      for (int b = 0; b < FB; ++b) in.z0[FB + 1 + b] = ld4(a.agg + row * F + b * 16 + 4 * kg);
    } else {
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) in.z0[nt] = ld4(a.z1 + row * H1 + nt * 16 + 4 * kg);
    }
  };
  auto compute = [&](int t, const MlpFwdIn<F>& in) {
    const bool valid = t * 16 + j < a.n_idx;
    const int64_t row = in.row;
    const int64_t srow = (int64_t)slot * a.srow_stride + a.idx_base + min(t * 16 + j, a.n_idx - 1);
    // ---- Dense 0: [2F+9] -> 80, relu
    f32x4 z1[1][5];
    if constexpr (F > 0) {
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) z1[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB1; ++kb) {
        f32x4 blk[1] = {in.z0[kb]};
        mfma_cols<1, 5>(smem + L::W1, LD1, kb, j, kg, blk, z1);
      }
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) {
        z1[0][nt] = relu4(z1[0][nt] + ld4(smem + L::B1 + nt * 16 + 4 * kg));
        if (valid) st4(a.z1 + row * H1 + nt * 16 + 4 * kg, z1[0][nt]);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) z1[0][nt] = in.z0[nt];
    }
    // ---- Dense 1: 80 -> 40 (48 padded), relu
    f32x4 z2[1][3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) z2[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      f32x4 blk[1] = {z1[0][kb]};
      mfma_cols<1, 3>(smem + L::W2, LD2, kb, j, kg, blk, z2);
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      z2[0][nt] = relu4(z2[0][nt] + ld4(smem + L::B2 + nt * 16 + 4 * kg));
      if (valid && nt * 16 + 4 * kg < H2) st4(a.z2 + srow * H2 + nt * 16 + 4 * kg, z2[0][nt]);
    }
    // ---- Dense 2: 40 -> 20 (32 padded), relu
    f32x4 z3[1][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) z3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      f32x4 blk[1] = {z2[0][kb]};
      mfma_cols<1, 2>(smem + L::W3, LD3, kb, j, kg, blk, z3);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      z3[0][nt] = relu4(z3[0][nt] + ld4(smem + L::B3 + nt * 16 + 4 * kg));
      if (valid && nt * 16 + 4 * kg < H3) st4(a.z3 + srow * H3 + nt * 16 + 4 * kg, z3[0][nt]);
    }
    // ---- Dense 3: 20 -> C (16 padded), linear
    f32x4 qa[1][1];
    qa[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 blk[1] = {z3[0][kb]};
      mfma_cols<1, 1>(smem + L::W4, LD4, kb, j, kg, blk, qa);
    }
    const f32x4 v = qa[0][0] + ld4(smem + L::B4 + 4 * kg);
    if (valid && 4 * kg < a.C) st4(a.q + row * a.C + 4 * kg, v);
  };

  MlpFwdIn<F> inA, inB;
  V2X_TILE_PROLOGUE(blockIdx.x * per + wv, t_end, inA, load_in)
  mlp_fill_lds<F>(smem, a, slot, true, threadIdx.x);
  __syncthreads();
  V2X_TILE_LOOP(inA, inB, load_in, compute)
}

template <int F>
struct MlpBwdIn { f32x4 q, y, z3[2], z2[3], z1[5]; int64_t row; };

// Huber (delta = 1, tf.losses.huber_loss BS_brain.py:86-87) + reverse chain through the MLP.
// Writes the pre-activation gradients dq, dz3, dz2, dz1 (for k_wgrad) and [dh | dagg].  Persistent like k_mlp_fwd.
template <int F>
__global__ __launch_bounds__(256, 2) void k_mlp_bwd(MlpArgs a) {
  using L = MlpLds<F>;
  constexpr int FB = F / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int slot = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int n_tiles = (a.n_idx + 15) >> 4;
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int t_end = min((int)(blockIdx.x + 1) * per, n_tiles);
  // lanes whose 4 columns lie in the zero padding of a narrow layer read column 0 instead (unconditional
  // loads; the value only gates a gradient that is already zero there)
  const int c3[2] = {4 * kg, 16 + 4 * kg < H3 ? 16 + 4 * kg : 0};
  const int c2[3] = {4 * kg, 16 + 4 * kg, 32 + 4 * kg < H2 ? 32 + 4 * kg : 0};
  const int cq = 4 * kg < a.C ? 4 * kg : 0;

  auto load_in = [&](int t, MlpBwdIn<F>& in) {
    const int idx = min(t * 16 + j, a.n_idx - 1);
    const int64_t row = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
    in.row = row;
    const int64_t srow = (int64_t)slot * a.srow_stride + a.idx_base + idx;
    in.q = ld4(a.q + row * a.C + cq);
    in.y = ld4(a.y + row * a.C + cq);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) in.z3[nt] = ld4(a.z3 + srow * H3 + c3[nt]);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) in.z2[nt] = ld4(a.z2 + srow * H2 + c2[nt]);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) in.z1[nt] = ld4(a.z1 + row * H1 + nt * 16 + 4 * kg);
  };
  auto compute = [&](int t, const MlpBwdIn<F>& in) {
    const bool valid = t * 16 + j < a.n_idx;
    const int64_t row = in.row;
    const int64_t srow = (int64_t)slot * a.srow_stride + a.idx_base + min(t * 16 + j, a.n_idx - 1);
    f32x4 g4[1];        // dq block (16 wide, only channels < C non-zero)
    g4[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (4 * kg < a.C) {
      float ls = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float err = in.q[c] - in.y[c];
        const float ab = fabsf(err), quad = fminf(ab, 1.f);
        ls += 0.5f * quad * quad + (ab - quad);
        g4[0][c] = fminf(fmaxf(err, -1.f), 1.f) * a.inv_denom;
      }
      if (valid) {
        st4(a.dq + srow * a.C + 4 * kg, g4[0]);
        a.rowloss[srow] = ls;      // C == 4: exactly one lane (kg == 0) per row
      }
    }
    // ---- Dense 3 backward: dz3 = (dq . W4^T) * (z3 > 0)
    const auto lin = [](int nt) { return nt * 16; };
    f32x4 d3[2][1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) d3[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows_multi<2>(smem + L::W4, LD4, lin, 0, j, kg, g4[0], d3);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const bool in_range = nt * 16 + 4 * kg < H3;
      d3[nt][0] = in_range ? gate4(d3[nt][0], in.z3[nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (valid && in_range) st4(a.dz3 + srow * H3 + nt * 16 + 4 * kg, d3[nt][0]);
    }
    // ---- Dense 2 backward: dz2 = (dz3 . W3^T) * (z2 > 0)
    f32x4 d2[3][1];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) mfma_rows_multi<3>(smem + L::W3, LD3, lin, kb, j, kg, d3[kb][0], d2);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const bool in_range = nt * 16 + 4 * kg < H2;
      d2[nt][0] = in_range ? gate4(d2[nt][0], in.z2[nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (valid && in_range) st4(a.dz2 + srow * H2 + nt * 16 + 4 * kg, d2[nt][0]);
    }
    // ---- Dense 1 backward: dz1 = (dz2 . W2^T) * (z1 > 0)
    f32x4 d1[5][1];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) mfma_rows_multi<5>(smem + L::W2, LD2, lin, kb, j, kg, d2[kb][0], d1);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      d1[nt][0] = gate4(d1[nt][0], in.z1[nt]);
      if (valid) st4(a.dz1 + row * H1 + nt * 16 + 4 * kg, d1[nt][0]);
    }
    // ---- Dense 0 backward (data): [dh | dagg] = dz1 . W1^T   (h rows and agg rows of the image)
    if constexpr (F > 0) {
      const auto skip_xe = [](int nt) { return nt < FB ? nt * 16 : F + XE + (nt - FB) * 16; };
      f32x4 o[2 * FB][1];
#pragma unroll
      for (int nt = 0; nt < 2 * FB; ++nt) o[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 5; ++kb) mfma_rows_multi<2 * FB>(smem + L::W1, LD1, skip_xe, kb, j, kg, d1[kb][0], o);
#pragma unroll
      for (int nt = 0; nt < 2 * FB; ++nt)
        if (valid) st4(a.gha + row * (2 * F) + nt * 16 + 4 * kg, o[nt][0]);
    }
  };

  MlpBwdIn<F> inA, inB;
  V2X_TILE_PROLOGUE(blockIdx.x * per + wv, t_end, inA, load_in)
  mlp_fill_lds<F>(smem, a, slot, false, threadIdx.x);
  __syncthreads();
  V2X_TILE_LOOP(inA, inB, load_in, compute)
}

// Training form: forward + Huber + reverse chain of the decision MLP in ONE pass over the rows (the hidden
// activations never leave the registers between the two directions; z1..z3 and the pre-activation gradients are
// still written once for k_wgrad).  Saves a launch and the read-back of q, z1, z2, z3 per fit step.
template <int F>
struct MlpTrainIn { f32x4 z0[2 * (F / 16) + 1]; f32x4 y; int64_t row; };

template <int F>
__global__ __launch_bounds__(256, 2) void k_mlp_train(MlpArgs a) {
  using L = MlpLds<F>;
  constexpr int FB = F / 16, KB1 = 2 * FB + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int slot = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int n_tiles = (a.n_idx + 15) >> 4;
  const int per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const int t_end = min((int)(blockIdx.x + 1) * per, n_tiles);
  const int cq = 4 * kg < a.C ? 4 * kg : 0;

  auto load_in = [&](int t, MlpTrainIn<F>& in) {
    const int idx = min(t * 16 + j, a.n_idx - 1);
    const int64_t row = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
    in.row = row;
#pragma unroll
    for (int b = 0; b < FB; ++b) in.z0[b] = ld4(a.h + row * F + b * 16 + 4 * kg);
    in.z0[FB] = ld4(a.xe + row * XE + 4 * kg);
#pragma unroll
    for (int b = 0; b < FB; ++b) in.z0[FB + 1 + b] = ld4(a.agg + row * F + b * 16 + 4 * kg);
    in.y = ld4(a.y + row * a.C + cq);
  };
  auto compute = [&](int t, const MlpTrainIn<F>& in) {
    const bool valid = t * 16 + j < a.n_idx;
    const int64_t row = in.row;
    const int64_t srow = (int64_t)slot * a.srow_stride + a.idx_base + min(t * 16 + j, a.n_idx - 1);
    // ================= forward
    f32x4 z1[1][5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) z1[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB1; ++kb) {
      f32x4 blk[1] = {in.z0[kb]};
      mfma_cols<1, 5>(smem + L::W1, LD1, kb, j, kg, blk, z1);
    }
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      z1[0][nt] = relu4(z1[0][nt] + ld4(smem + L::B1 + nt * 16 + 4 * kg));
      if (valid) st4(a.z1 + row * H1 + nt * 16 + 4 * kg, z1[0][nt]);
    }
    f32x4 z2[1][3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) z2[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      f32x4 blk[1] = {z1[0][kb]};
      mfma_cols<1, 3>(smem + L::W2, LD2, kb, j, kg, blk, z2);
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      z2[0][nt] = relu4(z2[0][nt] + ld4(smem + L::B2 + nt * 16 + 4 * kg));
      if (valid && nt * 16 + 4 * kg < H2) st4(a.z2 + srow * H2 + nt * 16 + 4 * kg, z2[0][nt]);
    }
    f32x4 z3[1][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) z3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      f32x4 blk[1] = {z2[0][kb]};
      mfma_cols<1, 2>(smem + L::W3, LD3, kb, j, kg, blk, z3);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      z3[0][nt] = relu4(z3[0][nt] + ld4(smem + L::B3 + nt * 16 + 4 * kg));
      if (valid && nt * 16 + 4 * kg < H3) st4(a.z3 + srow * H3 + nt * 16 + 4 * kg, z3[0][nt]);
    }
    f32x4 qa[1][1];
    qa[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 blk[1] = {z3[0][kb]};
      mfma_cols<1, 1>(smem + L::W4, LD4, kb, j, kg, blk, qa);
    }
    const f32x4 qv = qa[0][0] + ld4(smem + L::B4 + 4 * kg);
    if (valid && 4 * kg < a.C) st4(a.q + row * a.C + 4 * kg, qv);
    // ================= Huber (delta = 1) and the reverse chain
    f32x4 g4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (4 * kg < a.C) {
      float ls = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float err = qv[c] - in.y[c];
        const float ab = fabsf(err), quad = fminf(ab, 1.f);
        ls += 0.5f * quad * quad + (ab - quad);
        g4[c] = fminf(fmaxf(err, -1.f), 1.f) * a.inv_denom;
      }
      if (valid) {
        st4(a.dq + srow * a.C + 4 * kg, g4);
        a.rowloss[srow] = ls;
      }
    }
    const auto lin = [](int nt) { return nt * 16; };
    f32x4 d3[2][1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) d3[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows_multi<2>(smem + L::W4, LD4, lin, 0, j, kg, g4, d3);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const bool in_range = nt * 16 + 4 * kg < H3;
      d3[nt][0] = in_range ? gate4(d3[nt][0], z3[0][nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (valid && in_range) st4(a.dz3 + srow * H3 + nt * 16 + 4 * kg, d3[nt][0]);
    }
    f32x4 d2[3][1];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) mfma_rows_multi<3>(smem + L::W3, LD3, lin, kb, j, kg, d3[kb][0], d2);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const bool in_range = nt * 16 + 4 * kg < H2;
      d2[nt][0] = in_range ? gate4(d2[nt][0], z2[0][nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (valid && in_range) st4(a.dz2 + srow * H2 + nt * 16 + 4 * kg, d2[nt][0]);
    }
    f32x4 d1[5][1];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) mfma_rows_multi<5>(smem + L::W2, LD2, lin, kb, j, kg, d2[kb][0], d1);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      d1[nt][0] = gate4(d1[nt][0], z1[0][nt]);
      if (valid) st4(a.dz1 + row * H1 + nt * 16 + 4 * kg, d1[nt][0]);
    }
    const auto skip_xe = [](int nt) { return nt < FB ? nt * 16 : F + XE + (nt - FB) * 16; };
    f32x4 o[2 * FB][1];
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt) o[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) mfma_rows_multi<2 * FB>(smem + L::W1, LD1, skip_xe, kb, j, kg, d1[kb][0], o);
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt)
      if (valid) st4(a.gha + row * (2 * F) + nt * 16 + 4 * kg, o[nt][0]);
  };

  MlpTrainIn<F> inA, inB;
  V2X_TILE_PROLOGUE(blockIdx.x * per + wv, t_end, inA, load_in)
  mlp_fill_lds<F>(smem, a, slot, true, threadIdx.x);
  __syncthreads();
  V2X_TILE_LOOP(inA, inB, load_in, compute)
}

// =====================================================================================
// k_wgrad : dW[k][n] = sum_rows in[row][k] * dpre[row][n],  db[n] = sum_rows dpre[row][n]
//           per (row-chunk, slot) partial written to a slab; k_reduce_adam sums the slabs
// =====================================================================================
struct WgSeg { const float* ptr; int stride; int width; int col; int slot_major; };   // width real cols, col = padded K offset;
                                                                                     // slot_major: rows stored slot * srow_stride + idx (MlpArgs)
struct WgradArgs {
  WgSeg seg[3]; int n_seg;
  const float* dpre; int d_stride; int n_real;    // dpre[R][n_real]
  int kp, np;                                     // padded K and N_out (multiples of 16)
  RowPad pad;                                     // padded K row -> real row of the weight
  float* slab; int64_t slab_stride;               // slab[chunk][P]
  int64_t layer_off; int64_t slot_stride;         // where this layer's (slot 0) block starts in P
  int n_idx, row_stride, base_mul, chunk;         // idx range of chunk c: [c*chunk, (c+1)*chunk)
  const float* zeros;                             // >= 16 B of zeros (source of absent segments)
  int idx_base, chunk_base;                       // sub-range of the batch; first slab index of this launch
  int n_chunks;                                   // workgroups (slabs) of THIS role; blockIdx.x >= n_chunks exits
  int kind;                                       // WG_KIND_*: selects the compile-time operand widths
  int dpre_slot_major; int srow_stride;           // row layout of dpre / of slot-major segments
  long long* ts;                                  // phase stamps (diagnostics, V2X_FUSED_TS=1) or null
  // the embed layer's gradient riding on a graph layer's role (wgrad_body EN > 0): dW0[x|e rows][e_col0 .. + 16 EN) =
  // xe^T . dpre_0[:, those columns], written to the embed layer's block of the same slab (its neighbour-init rows: zeros)
  const float* e_dpre; int e_stride, e_col0, e_n_real;
  int64_t e_layer_off, e_slot_stride; RowPad e_pad;
  // wgrad_body FRAGK: the K0 and K2 segments are FRAGMENT-major (MlpArgs::frag_groups: h_L and a_L as the fused graph-layer
  // kernels leave them for k_mlp_train_wg), this many 16-row groups per slot
  int frag_groups;
  // a role that covers only PART of the layer's K rows (Dense-0 cut in two, WG_KIND_DENSE0A / B): padded K row of its first
  // segment; the bias gradient is written by the role with bias_too set
  int k_off, no_bias;
  // k_wgrad MODE 4: the next `chain` roles of the launch have no workgroups of their own -- this role's workgroup runs them after
  // its own body, for the same (chunk, slot): the light Dense 1..3 roles behind the Dense-0 halves, whose workgroups are the
  // lightest of the grid (25 + 8 and 20 + 15 accumulator tiles against a graph layer's 36 + its share of the embed layer)
  int chain, pad_;
};

constexpr int WG_TR = 16;        // rows per MFMA block (chunk sizes are multiples of it)

// ---- operand of width W (multiple of 16) in MFMA fragment layout, straight from HBM -------------------------
// The assignment of features to (tile, lane) inside an operand is free, so every full group of 64 features is
// labelled   feature = 64*g + 4*lane + t   (t = tile within the group): ONE float4 load per lane and k-step (a
// full 256-byte row per 16 lanes) feeds FOUR tiles.  The remaining 16-wide tiles use feature = 16*r + lane (dword
// loads).  A 144-wide GNN input is 3 loads per k-step instead of 9, and two blocks in flight stay below the
// 6-bit vmcnt counter (above it hipcc falls back to vmcnt(0) and the prefetch is lost).
template <int W>
struct WgOperand {
  static constexpr int G = W / 64;              // float4 groups
  static constexpr int R = (W % 64) / 16;       // dword tiles
  static constexpr int T = W / 16;              // MFMA tiles
  f32x4 g[G > 0 ? G : 1][4];                    // [group][k-step]
  float d[R > 0 ? R : 1][4];                    // [tile][k-step]
  // value of tile q at k-step s
  __device__ __forceinline__ float get(int q, int s) const {
    if (q < 4 * G) {
      return g[q >> 2][s][q & 3];
    }
    return d[q - 4 * G][s];
  }
  // feature (column inside the operand) that (tile q, lane index i) stands for
  static __device__ __forceinline__ int feature(int q, int i) {
    return q < 4 * G ? 64 * (q >> 2) + 4 * i + (q & 3) : 64 * G + 16 * (q - 4 * G) + i;
  }
};

// source of one operand: wave-uniform base + row stride, real width (columns >= width are clamped: the dW entries
// they feed are never written)
struct WgSrc { gfloat_p p; unsigned stride; int width; unsigned rs, off; };   // row(idx) = idx * rs + off

template <int W>
__device__ __forceinline__ void wg_load(WgOperand<W>& o, const WgSrc& src, unsigned row, int s, int j, float mk) {
  if constexpr (W > 0) {
    const unsigned base = row * src.stride;
#pragma unroll
    for (int g = 0; g < WgOperand<W>::G; ++g) {      // widths that are multiples of 64 are always complete
      typedef const __attribute__((address_space(1))) f32x4* gvec_p;
      const f32x4 v = *(gvec_p)(src.p + base + 64 * g + 4 * j);
      o.g[g][s] = v * mk;
    }
#pragma unroll
    for (int r = 0; r < WgOperand<W>::R; ++r) {
      const int c = min(64 * WgOperand<W>::G + 16 * r + j, src.width - 1);
      o.d[r][s] = src.p[base + c] * mk;
    }
  }
}

// The same operand from a FRAGMENT-major source (MlpArgs::frag_groups): per (slot, 16-row group) W/16 blocks of 256 floats,
// block b = features 16b .. 16b+15, element (row r, feature f) at ((f & 15) >> 2) * 64 + r * 4 + (f & 3).  With the labelling
// feature = 64 g + 4 j + t the four tiles t of lane index j are one float4 again: block 4 g + (j >> 2), row group j & 3.
// tile: element offset of the (slot, group) tile; r: the lane's row inside the tile at this k-step (4 kg + s).
template <int W>
__device__ __forceinline__ void wg_load_frag(WgOperand<W>& o, gfloat_p p, unsigned tile, int r, int s, int j) {
  static_assert(W % 64 == 0, "fragment-major operands are whole 64-feature groups");
#pragma unroll
  for (int g = 0; g < WgOperand<W>::G; ++g) {
    typedef const __attribute__((address_space(1))) f32x4* gvec_p;
    o.g[g][s] = *(gvec_p)(p + tile + (4 * g + (j >> 2)) * 256 + ((j & 3) * 16 + r) * 4);
  }
}

// K operands K0|K1|K2 (compile-time padded widths, 0 = absent), N operand NW.  Every wave owns ALL output tiles for
// ITS OWN 16-row blocks (blocks wv, wv+4, ... of the chunk): each element is loaded from HBM exactly once, 4*KT*NT
// MFMAs per block (4.6k cycles for a GNN stage) run while the next block's loads are in flight (register double
// buffer, branch-free steady state, counted vmcnt), no LDS or barrier in the loop; the four accumulator sets are
// summed through LDS once at the end and written as float4.  (Measured alternatives, all ~2x slower: LDS-transposed
// tiles with a barrier per tile; a 2 x 2 tile split over the waves that loads every operand twice; dword-only
// fragment loads; single buffering at 4 waves/SIMD -- waves sharing a SIMD run in lockstep, so occupancy alone
// does not overlap memory and MFMA phases.)
// ZERO1: the K1 segment is absent (identically zero input, e.g. the embed layer's neighbour-init): nothing is loaded or
// multiplied for its tiles, their gradient rows are written as exact zeros.
// EN > 0: this role also produces EN output tiles of the EMBED layer's gradient (K1 must be the [x|e] tile): the light
// embed role (16 MFMAs per block, load-bound) otherwise needs workgroups of its own, which queue behind the heavy ones.
// FRAGK: K0 and K2 come from fragment-major buffers (Dense-0's h_L / a_L behind the fused graph-layer kernels: the role the
// decision-MLP launch hands to this one at small batches, kernels_mlpwg.hpp WG0 = false); whole 16-row groups only.
template <int K0, int K1, int K2, int NW, int DEPTH = 2, bool ZERO1 = false, int EN = 0, bool FRAGK = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, float* smem, const int bx, const int slot) {
  constexpr int T0 = K0 / 16, T1 = K1 / 16, T2 = K2 / 16, KT = T0 + T1 + T2, NT = NW / 16;
  constexpr int EW = 16 * EN, ENA = EN > 0 ? EN : 1;
  static_assert(EN == 0 || (T1 == 1 && !ZERO1), "the embed gradient rides on the [x|e] tile");
  if (bx >= a.n_chunks) return;                                  // roles have work-proportional grids
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform
  const int j = lane & 15, kg = lane >> 4;
  const int i_begin = bx * a.chunk, i_end = min(i_begin + a.chunk, a.n_idx);
  long long* tsp = (a.ts && bx == 1 && slot == 0 && lane == 0) ? a.ts + wv * 64 : nullptr;      // (ts: one role of the launch only)
  int tsn = 0;
  auto mark = [&]() {
    if (tsp && tsn < 64) tsp[tsn] = tsn == 0 ? (long long)wall_clock64() : (long long)__builtin_readcyclecounter();
    ++tsn;
  };
  mark(); mark();

  // (the descriptor was copied out of the kernarg segment as raw words: cast to the global address space
  //  explicitly, else hipcc emits flat loads and waits vmcnt(0) lgkmcnt(0) before every MFMA)
  WgSrc src[3], srcn;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool have = i < a.n_seg && a.seg[i].ptr != nullptr;
    src[i].p = (gfloat_p)(have ? a.seg[i].ptr : a.zeros);
    src[i].stride = have ? a.seg[i].stride : 0;
    src[i].width = have ? a.seg[i].width : 64;               // zero buffer: 64 floats, stride 0
    const bool sm = have && a.seg[i].slot_major;
    src[i].rs = sm ? 1u : (unsigned)a.row_stride;
    src[i].off = sm ? (unsigned)slot * (unsigned)a.srow_stride : (unsigned)slot * (unsigned)a.base_mul;
  }
  srcn.p = (gfloat_p)a.dpre; srcn.stride = a.d_stride; srcn.width = a.n_real;
  srcn.rs = a.dpre_slot_major ? 1u : (unsigned)a.row_stride;
  srcn.off = a.dpre_slot_major ? (unsigned)slot * (unsigned)a.srow_stride : (unsigned)slot * (unsigned)a.base_mul;
  WgSrc srce;
  srce.p = (gfloat_p)(EN > 0 ? a.e_dpre + a.e_col0 : a.zeros); srce.stride = EN > 0 ? a.e_stride : 0; srce.width = EW > 0 ? EW : 16;
  srce.rs = (unsigned)a.row_stride; srce.off = (unsigned)slot * (unsigned)a.base_mul;
  f32x4 acce[ENA];
  float bsume[ENA];
#pragma unroll
  for (int e = 0; e < ENA; ++e) { acce[e] = (f32x4){0.f, 0.f, 0.f, 0.f}; bsume[e] = 0.f; }

  f32x4 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  constexpr int K1L = ZERO1 ? 0 : K1;                            // loaded width of the K1 operand
  struct Block { WgOperand<K0> k0; WgOperand<K1L> k1; WgOperand<K2> k2; WgOperand<NW> n; WgOperand<EW> e; };
  auto kval = [&](const Block& b, int kt, int s) -> float {
    if (kt < T0) return b.k0.get(kt, s);
    if (kt < T0 + T1) return ZERO1 ? 0.f : b.k1.get(kt - T0, s);
    return b.k2.get(kt - T0 - T1, s);
  };
  auto mfma_block = [&](const Block& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (ZERO1 && kt >= T0 && kt < T0 + T1) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = V2X_MFMA(kval(b, kt, s), b.n.get(nt, s), acc[kt][nt]);
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bsum[nt] += (b.n.get(nt, 0) + b.n.get(nt, 1)) + (b.n.get(nt, 2) + b.n.get(nt, 3));
    if constexpr (EN > 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < EN; ++e) acce[e] = V2X_MFMA(kval(b, T0, s), b.e.get(e, s), acce[e]);
#pragma unroll
      for (int e = 0; e < EN; ++e) bsume[e] += (b.e.get(e, 0) + b.e.get(e, 1)) + (b.e.get(e, 2) + b.e.get(e, 3));
    }
  };
  const int n_rows_here = max(i_end - i_begin, 0);
  const int n_full = n_rows_here / WG_TR;
  const unsigned idx_lane0 = (unsigned)(a.idx_base + i_begin + 4 * kg);
  const unsigned ftile0 = (unsigned)slot * (unsigned)a.frag_groups + (unsigned)((a.idx_base + i_begin) >> 4);   // FRAGK
  auto load_full = [&](int blk, Block& b) {                      // full block: no clamps on rows, no masks
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned idx = idx_lane0 + (unsigned)(blk * WG_TR + s);
      if constexpr (FRAGK) wg_load_frag<K0>(b.k0, src[0].p, (ftile0 + (unsigned)blk) * (unsigned)(K0 * 16), 4 * kg + s, s, j);
      else wg_load<K0>(b.k0, src[0], idx * src[0].rs + src[0].off, s, j, 1.f);
      wg_load<K1L>(b.k1, src[1], idx * src[1].rs + src[1].off, s, j, 1.f);
      if constexpr (FRAGK) wg_load_frag<K2>(b.k2, src[2].p, (ftile0 + (unsigned)blk) * (unsigned)(K2 * 16), 4 * kg + s, s, j);
      else wg_load<K2>(b.k2, src[2], idx * src[2].rs + src[2].off, s, j, 1.f);
      wg_load<NW>(b.n, srcn, idx * srcn.rs + srcn.off, s, j, 1.f);
      wg_load<EW>(b.e, srce, idx * srce.rs + srce.off, s, j, 1.f);
    }
  };

  // This wave's blocks: wv, wv+4, ...  The loads of the next block are issued, THEN the MFMAs of the current one
  // run (sched_barrier pins that order; a load under an `if` would force vmcnt(0) at the join).
  const int nb = n_full > wv ? (n_full - wv + 3) >> 2 : 0;
  // Three register buffers, TWO blocks of loads in flight while a third is multiplied: the launch is bound by HBM
  // latency x bytes in flight per CU (324 MB per step through ~10 B/clk/CU) at least as much as by the MFMA pipe,
  // and one wave per SIMD (512-VGPR budget) is the only place to hold more bytes in flight.
  // (DEPTH 3 where the operand buffers fit the 256 architectural VGPRs next to the accumulators; DEPTH 2 otherwise)
  Block b0, b1;
#define V2X_SB __builtin_amdgcn_sched_barrier(0)
  // A wave issues in order: a block's loads issued as one burst in front of the MFMAs keep the matrix pipe idle while the
  // texture unit takes them (tools/l2stream.hip, tools/overlapbench.hip).  The loads of the NEXT block are therefore
  // spread between the MFMAs of the current one, one per PER MFMAs.
  constexpr int NLD = 4 * ((K0 > 0 ? WgOperand<K0>::G + WgOperand<K0>::R : 0) + (K1L > 0 ? WgOperand<K1L>::G + WgOperand<K1L>::R : 0) +
                           (K2 > 0 ? WgOperand<K2>::G + WgOperand<K2>::R : 0) + WgOperand<NW>::G + WgOperand<NW>::R +
                           (EW > 0 ? WgOperand<EW>::G + WgOperand<EW>::R : 0));
  constexpr int NMF = 4 * (KT - (ZERO1 ? T1 : 0)) * NT + 4 * EN;
  constexpr int PER = NMF / NLD > 0 ? NMF / NLD : 1;
#define V2X_ILV                                                                                        \
  {                                                                                                    \
    _Pragma("unroll") for (int u_ = 0; u_ < NLD; ++u_) {                                               \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                                             \
    }                                                                                                  \
  }                                                                                                    \
  V2X_SB
  int k = 0;
  if constexpr (DEPTH == 2) {
    if (nb > 0) load_full(wv, b0);
#pragma unroll 1
    for (; k + 2 < nb; k += 2) {
      load_full(wv + 4 * (k + 1), b1);
      mfma_block(b0); V2X_ILV;
      mark();
      load_full(wv + 4 * (k + 2), b0);
      mfma_block(b1); V2X_ILV;
      mark();
    }
    if (nb - k == 2) {                                             // peeled tail: 2 or 1 blocks left
      load_full(wv + 4 * (k + 1), b1); V2X_SB;
      mfma_block(b0);
      mfma_block(b1);
    } else if (nb - k == 1) {
      mfma_block(b0);
    }
  } else if (nb > 0) {
    // DEPTH register buffers, DEPTH - 1 blocks of loads in flight while one is multiplied.  Block k + u of a group of
    // DEPTH lives in buffer u; loads past the wave's last block re-request that block (unconditional: a load under an
    // `if` would force vmcnt(0) at the join; the repeats hit the L1 / L2)
    Block br[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load_full(wv + 4 * min(d, nb - 1), br[d]);
    V2X_SB;
#pragma unroll 1
    for (; k + DEPTH <= nb; k += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        load_full(wv + 4 * min(k + u + DEPTH - 1, nb - 1), br[(u + DEPTH - 1) % DEPTH]);
        mfma_block(br[u]); V2X_ILV;
        mark();
      }
    }
    const int rem = nb - k;                  // 0 .. DEPTH - 1 blocks left, already requested, in buffers 0 .. rem - 1
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u)
      if (u < rem) mfma_block(br[u]);
  }
#undef V2X_ILV
#undef V2X_SB
  if (n_full * WG_TR < n_rows_here && wv == (n_full & 3)) {      // partial last block of the chunk
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ridx = i_begin + n_full * WG_TR + 4 * kg + s;
      const float mk = ridx < i_end ? 1.f : 0.f;                 // rows past the chunk contribute 0
      const unsigned idx = (unsigned)(a.idx_base + min(ridx, a.n_idx - 1));
      wg_load<K0>(b0.k0, src[0], idx * src[0].rs + src[0].off, s, j, 1.f);
      wg_load<K1L>(b0.k1, src[1], idx * src[1].rs + src[1].off, s, j, 1.f);
      wg_load<K2>(b0.k2, src[2], idx * src[2].rs + src[2].off, s, j, 1.f);
      wg_load<NW>(b0.n, srcn, idx * srcn.rs + srcn.off, s, j, mk);
      wg_load<EW>(b0.e, srce, idx * srce.rs + srce.off, s, j, mk);
    }
    mfma_block(b0);
  }

  // ---- sum the 4 waves' accumulators through LDS as (w0 + w1) + (w2 + w3): two sets (one per wave pair); in a pair
  //      one wave STORES a tile and the other adds its own in place -- the even wave stores the first half of the tiles
  //      and adds the second, the odd wave the other way round (a + b == b + a bitwise), so both work in both rounds; the
  //      writer below adds the two sets.  Fixed order => deterministic.  (Until round 2 this was "wave 0 stores, 1..3
  //      add in turn": three rounds of dependent LDS read-modify-writes by ONE wave, measured ~0.4 us per tile.)
  //      Lane (kg, j) of tile (kt, nt) holds rows feature_k(kt, 4*kg + r), column feature_n(nt, j).
  mark();
  constexpr int TILES_X = KT * NT + EN, XHALF = TILES_X / 2, NTB = NT + EN;
  f32x4* sAcc = reinterpret_cast<f32x4*>(smem);                  // set A [KT*NT (+ EN)][64 lanes]
  f32x4* sAcc2 = sAcc + TILES_X * 64;                            // set B
  float* sBias4 = smem + 2 * TILES_X * 64 * 4;                   // [wave][NT (+ EN)][16]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {                              // bias: fold the 4 row groups of the wave
    float v = bsum[nt];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    bsum[nt] = v;
  }
  if constexpr (EN > 0) {
#pragma unroll
    for (int e = 0; e < EN; ++e) {
      float v = bsume[e];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      bsume[e] = v;
    }
  }
  if (kg == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) sBias4[(wv * NTB + nt) * 16 + j] = bsum[nt];
    if constexpr (EN > 0) {
#pragma unroll
      for (int e = 0; e < EN; ++e) sBias4[(wv * NTB + NT + e) * 16 + j] = bsume[e];
    }
  }
  {
    f32x4* sSet = (wv >> 1 ? sAcc2 : sAcc) + lane;
    const bool even = (wv & 1) == 0;
    auto xchg = [&](auto STORE, auto LOW) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int ti = kt * NT + nt;
          if ((ti < XHALF) == decltype(LOW)::value) {
            if (decltype(STORE)::value) sSet[ti * 64] = acc[kt][nt];
            else sSet[ti * 64] += acc[kt][nt];
            if ((ti & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
      if constexpr (EN > 0) {
#pragma unroll
        for (int e = 0; e < EN; ++e) {
          const int ti = KT * NT + e;
          if ((ti < XHALF) == decltype(LOW)::value) {
            if (decltype(STORE)::value) sSet[ti * 64] = acce[e];
            else sSet[ti * 64] += acce[e];
          }
        }
      }
    };
    if (even) xchg(std::true_type{}, std::true_type{}); else xchg(std::true_type{}, std::false_type{});
    __syncthreads();
    if (even) xchg(std::false_type{}, std::false_type{}); else xchg(std::false_type{}, std::true_type{});
    __syncthreads();
  }
  // ---- write the partial.  Padded K row of (kt, lane index i): operand column offset + feature(); output column of
  //      (nt, j): WgOperand<NW>::feature.  Inside a 64-wide N group the 4 tiles of a lane are 4 consecutive columns
  //      => float4 stores.
  float* dst = a.slab + (int64_t)(a.chunk_base + bx) * a.slab_stride + a.layer_off + slot * a.slot_stride;
  auto krow = [&](int kt, int i) -> int {
    if (kt < T0) return a.k_off + WgOperand<K0>::feature(kt, i);
    if (kt < T0 + T1) return a.k_off + K0 + WgOperand<K1>::feature(kt - T0, i);
    return a.k_off + K0 + K1 + WgOperand<K2>::feature(kt - T0 - T1, i);
  };
  constexpr int NG = WgOperand<NW>::G, NR = WgOperand<NW>::R;
  constexpr int UNITS = KT * (NG + NR);                          // (k tile, n group | n dword tile)
  for (int u = wv; u < UNITS; u += 4) {
    const int kt = u / (NG + NR), nn = u - kt * (NG + NR);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = real_row(a.pad, krow(kt, 4 * kg + r));
      if (rr < 0) continue;
      if (nn < NG) {                                             // 4 tiles -> 4 consecutive columns
        const int col = 64 * nn + 4 * j;
        float4 v;
        v.x = sAcc[(kt * NT + 4 * nn + 0) * 64 + lane][r] + sAcc2[(kt * NT + 4 * nn + 0) * 64 + lane][r];
        v.y = sAcc[(kt * NT + 4 * nn + 1) * 64 + lane][r] + sAcc2[(kt * NT + 4 * nn + 1) * 64 + lane][r];
        v.z = sAcc[(kt * NT + 4 * nn + 2) * 64 + lane][r] + sAcc2[(kt * NT + 4 * nn + 2) * 64 + lane][r];
        v.w = sAcc[(kt * NT + 4 * nn + 3) * 64 + lane][r] + sAcc2[(kt * NT + 4 * nn + 3) * 64 + lane][r];
        if (col + 3 < a.n_real) *reinterpret_cast<float4*>(dst + (int64_t)rr * a.n_real + col) = v;
      } else {
        const int nt = 4 * NG + (nn - NG), col = 64 * NG + 16 * (nn - NG) + j;
        if (col < a.n_real) dst[(int64_t)rr * a.n_real + col] = sAcc[(kt * NT + nt) * 64 + lane][r] + sAcc2[(kt * NT + nt) * 64 + lane][r];
      }
    }
  }
  if (tid < NT * 16 && !a.no_bias) {
    const int nt = tid >> 4, jj = tid & 15, col = WgOperand<NW>::feature(nt, jj);
    if (col < a.n_real)
      dst[(int64_t)a.pad.k_real * a.n_real + col] =
          (sBias4[tid] + sBias4[NTB * 16 + tid]) + (sBias4[2 * NTB * 16 + tid] + sBias4[3 * NTB * 16 + tid]);
  }
  if constexpr (EN > 0) {                                        // the embed layer's columns [e_col0, e_col0 + 16 EN)
    float* dste = a.slab + (int64_t)(a.chunk_base + bx) * a.slab_stride + a.e_layer_off + slot * a.e_slot_stride;
    const int nre = a.e_n_real;
    for (int e = wv; e < EN; e += 4) {                           // [x|e] rows x tile e
      const int ti = KT * NT + e, col = a.e_col0 + WgOperand<EW>::feature(e, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = real_row(a.e_pad, WgOperand<K1>::feature(0, 4 * kg + r));
        if (rr >= 0 && rr < a.e_pad.pad_at) dste[(int64_t)rr * nre + col] = sAcc[ti * 64 + lane][r] + sAcc2[ti * 64 + lane][r];
      }
    }
    // rows of the absent neighbour-init block (real rows pad_at .. k_real - 1): exact zeros
    for (int i = tid; i < (a.e_pad.k_real - a.e_pad.pad_at) * EW; i += 256) {
      const int rr = a.e_pad.pad_at + i / EW, c = i % EW;
      dste[(int64_t)rr * nre + a.e_col0 + c] = 0.f;
    }
    if (tid < EN * 16) {
      const int e = tid >> 4, jj = tid & 15, col = a.e_col0 + WgOperand<EW>::feature(e, jj), o = (NT + e) * 16 + jj;
      dste[(int64_t)a.e_pad.k_real * nre + col] = (sBias4[o] + sBias4[NTB * 16 + o]) + (sBias4[2 * NTB * 16 + o] + sBias4[3 * NTB * 16 + o]);
    }
  }
  mark();
  if (tsp && tsn < 64) tsp[tsn] = (long long)wall_clock64();
}

// Several independent weight-gradient problems (roles) in ONE launch: blockIdx.z selects the layer and each
// role runs the body instantiated for ITS operand widths; the kernel template only bounds the register budget.
// Every launch of this path costs ~6-8 us of fixed latency, so the 4 Dense layers (and the L+1 GNN stages)
// share one launch each instead of 4 (L+1).  Kernarg structs indexed by blockIdx.z would be copied to scratch
// (runtime-indexed array): the role's descriptor is read through the constant-address-space kernarg pointer.
constexpr int WG_MAX_ROLES = 8;
#ifndef V2X_WG_DEPTH_GNN
#define V2X_WG_DEPTH_GNN 3
#endif
// with the embed gradient on board (one more operand per block, 11 blocks per wave) two register buffers beat three:
// 42.3-43.5 us against 45.1 (depth 4: 45.9)
#ifndef V2X_WG_DEPTH_MERGED
#define V2X_WG_DEPTH_MERGED 2
#endif
#ifndef V2X_WG_DEPTH_D0
#define V2X_WG_DEPTH_D0 4
#endif
// packed: roles with DIFFERENT chunk counts in one grid.  A (chunks, slots, roles) grid sized for the largest count would launch
// workgroups that exit at once for the others, and workgroups go to the XCDs round-robin by linear id: with 4 chunks for the
// graph layers and 2 for the Dense-0 halves every real workgroup of the halves landed on XCDs 0, 1, 4, 5 -- 40 workgroups on 32
// CUs there, a second round, 30 us instead of 20 (round 6, profiles/r06_wgrad_roles_b512.txt).  Packed: a 1-D grid of exactly the
// real workgroups, role r owning ids [wg_begin[r], wg_begin[r + 1]), chunk fastest inside it.
struct WgradMulti { WgradArgs w[WG_MAX_ROLES]; int wg_begin[WG_MAX_ROLES + 1]; int packed; };
enum { WG_KIND_GNN = 0, WG_KIND_EMBED = 1, WG_KIND_DENSE0 = 2, WG_KIND_DENSE1 = 3, WG_KIND_DENSE2 = 4, WG_KIND_DENSE3 = 5,
       WG_KIND_EMBED_NONBR = 6, WG_KIND_GNN_E1 = 7, WG_KIND_GNN_E2 = 8, WG_KIND_GNN_E4 = 9,      // _En: + n tiles of the embed gradient
       // Dense-0 cut along K into [h | x] and [agg] (25 + 20 accumulator tiles instead of 45: the whole layer in one role spills),
       // _F: h_L / a_L fragment-major
       WG_KIND_DENSE0A = 10, WG_KIND_DENSE0A_F = 11, WG_KIND_DENSE0B = 12, WG_KIND_DENSE0B_F = 13 };

// MODE 0: the GNN stages, 1: the Dense layers, 2: both families in one launch (roles ordered heaviest first),
// 3: the GNN stages + Dense-0 in two halves (small batches: kernels_mlpwg.hpp WG0 = false leaves dz1 for it),
// 4: 3 + Dense 1..3 (smaller batches still: kernels_mlpstream.hpp)
template <int F, int MODE>
__global__ __launch_bounds__(256, 1) void k_wgrad(WgradMulti mu) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef const __attribute__((address_space(4))) unsigned* CWords;
  static_assert(sizeof(WgradArgs) % 4 == 0, "WgradArgs must be dword sized");
  constexpr int NW = sizeof(WgradArgs) / 4;
  CWords kw = (CWords)__builtin_amdgcn_kernarg_segment_ptr();
  int role = blockIdx.z, bx_ = blockIdx.x, slot_ = blockIdx.y;
  static_assert(offsetof(WgradMulti, wg_begin) == WG_MAX_ROLES * sizeof(WgradArgs), "wg_begin follows the roles");
  if (kw[WG_MAX_ROLES * NW + WG_MAX_ROLES + 1]) {                // packed (workgroup-uniform)
    const int id = blockIdx.x;
    role = 0;
#pragma unroll
    for (int r = 1; r < WG_MAX_ROLES; ++r) role += id >= (int)kw[WG_MAX_ROLES * NW + r] ? 1 : 0;   // (unused roles: begin = grid size)
    bx_ = id - (int)kw[WG_MAX_ROLES * NW + role];
  }
  CWords srcw = kw + role * NW;
  WgradArgs a;
  unsigned* dstw = reinterpret_cast<unsigned*>(&a);
#pragma unroll
  for (int i = 0; i < NW; ++i) dstw[i] = srcw[i];
  if (kw[WG_MAX_ROLES * NW + WG_MAX_ROLES + 1]) { slot_ = bx_ / a.n_chunks; bx_ -= slot_ * a.n_chunks; }
  if constexpr (MODE != 1) {
    if (a.kind == WG_KIND_GNN) { wgrad_body<F, XE, F, F, V2X_WG_DEPTH_GNN>(a, smem, bx_, slot_); return; }
    if (a.kind == WG_KIND_GNN_E1) { wgrad_body<F, XE, F, F, V2X_WG_DEPTH_MERGED, false, 1>(a, smem, bx_, slot_); return; }
    if constexpr (F >= 32) {
      if (a.kind == WG_KIND_GNN_E2) { wgrad_body<F, XE, F, F, V2X_WG_DEPTH_MERGED, false, 2>(a, smem, bx_, slot_); return; }
    }
    if constexpr (F >= 64) {
      if (a.kind == WG_KIND_GNN_E4) { wgrad_body<F, XE, F, F, V2X_WG_DEPTH_MERGED, false, 4>(a, smem, bx_, slot_); return; }
    }
    if (a.kind == WG_KIND_EMBED) { wgrad_body<XE, F, 0, F, 3>(a, smem, bx_, slot_); return; }
    if (a.kind == WG_KIND_EMBED_NONBR) { wgrad_body<XE, F, 0, F, 3, true>(a, smem, bx_, slot_); return; }
  }
  if constexpr (MODE == 3 || MODE == 4) {
    // (a half's block is 100 / 80 MFMAs, 1.2-1.4 us: with one block of loads in flight -- DEPTH 2 -- a wave waits 3.6 us per
    //  block for memory, measured at the 512- / 1024-graph shares; three in flight next to 100 accumulators still fit)
    if (a.kind == WG_KIND_DENSE0A) wgrad_body<F, XE, 0, H1, V2X_WG_DEPTH_D0>(a, smem, bx_, slot_);
    else if (a.kind == WG_KIND_DENSE0B) wgrad_body<F, 0, 0, H1, V2X_WG_DEPTH_D0>(a, smem, bx_, slot_);
    if constexpr (F == 64) {
      if (a.kind == WG_KIND_DENSE0A_F) wgrad_body<F, XE, 0, H1, V2X_WG_DEPTH_D0, false, 0, true>(a, smem, bx_, slot_);
      else if (a.kind == WG_KIND_DENSE0B_F) wgrad_body<F, 0, 0, H1, V2X_WG_DEPTH_D0, false, 0, true>(a, smem, bx_, slot_);
    }
  }
  if constexpr (MODE == 4) {              // + Dense 1..3 (kernels_mlpstream.hpp leaves their operands in memory), chained behind the halves
    const int n_chain = a.chain;          // (workgroup-uniform)
    for (int c = 1; c <= n_chain; ++c) {
      __syncthreads();                    // (the previous body's accumulator exchange is done with the LDS)
      CWords cw = kw + (role + c) * NW;
#pragma unroll
      for (int i = 0; i < NW; ++i) dstw[i] = cw[i];
      if (a.kind == WG_KIND_DENSE1) wgrad_body<H1, 0, 0, H2P>(a, smem, bx_, slot_);
      else if (a.kind == WG_KIND_DENSE2) wgrad_body<H2P, 0, 0, H3P>(a, smem, bx_, slot_);
      else if (a.kind == WG_KIND_DENSE3) wgrad_body<H3P, 0, 0, CP>(a, smem, bx_, slot_);
    }
  }
  if constexpr (MODE == 1 || MODE == 2) {
    if (a.kind == WG_KIND_DENSE0) wgrad_body<F, XE, F, H1>(a, smem, bx_, slot_);
    else if (a.kind == WG_KIND_DENSE1) wgrad_body<H1, 0, 0, H2P>(a, smem, bx_, slot_);
    else if (a.kind == WG_KIND_DENSE2) wgrad_body<H2P, 0, 0, H3P>(a, smem, bx_, slot_);
    else if (a.kind == WG_KIND_DENSE3) wgrad_body<H3P, 0, 0, CP>(a, smem, bx_, slot_);
  }
}

// =====================================================================================
// slab reduction + Keras Adam
// =====================================================================================
// Fragment-major copy of the GNN-layer weights for the graph-major fused kernels (kernels_fused.hpp, FzPack /
// k_pack_weights): the Adam kernel writes every updated parameter to its place in that copy as well, so that no
// re-packing launch is needed between two training steps.
struct AdamPack { float* fwd; float* bwd; int F, S, L, xr; };

// e = offset of a float4 of parameters in the flat buffer (GNN layers come first: stage-major, slot-minor)
__device__ __forceinline__ void pack_scatter(const AdamPack& k, int64_t e, float4 p) {
  const int F = k.F, FB = F >> 4, xr = k.xr;
  const int st0 = (xr + F) * F + F, st1 = (2 * F + xr) * F + F;          // floats per slot: stage 0, stages >= 1
  const int64_t size0 = (int64_t)k.S * st0;
  int stage, slot, within;
  if (e < size0) {
    stage = 0; slot = (int)(e / st0); within = (int)(e - (int64_t)slot * st0);
  } else {
    const int64_t t = e - size0, per = (int64_t)k.S * st1;
    stage = 1 + (int)(t / per);
    if (stage > k.L) return;                                             // Dense layers: not packed
    const int64_t r = t - (int64_t)(stage - 1) * per;
    slot = (int)(r / st1); within = (int)(r - (int64_t)slot * st1);
  }
  const int k_real = stage ? 2 * F + xr : xr + F, kbs = stage ? 2 * FB + 1 : 1;
  const int fwd_sz = kbs * FB * 256 + F;
  float* df = k.fwd + (stage ? (int64_t)k.S * (FB * 256 + F) + ((int64_t)(stage - 1) * k.S + slot) * fwd_sz : (int64_t)slot * fwd_sz);
  const int rr = within / F, col = within - rr * F;
  if (rr >= k_real) {                                                    // the bias row
    *reinterpret_cast<float4*>(df + kbs * FB * 256 + col) = p;
    return;
  }
  const int pad_at = stage ? F + xr : xr;
  const int kp = rr < pad_at ? rr : rr + (XE - xr);                      // row of the padded K axis
  if (stage || kp < 16) {                                                // (stage 0: only the xe block is used)
    const int kb = kp >> 4, kgp = (kp & 15) >> 2, s4 = kp & 3;
    const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = col + t;
      df[(((kb * FB + (n >> 4)) * 64) + (n & 15) + 16 * kgp) * 4 + s4] = pv[t];
    }
  }
  if (stage) {
    int ntb = -1;
    if (kp < F) ntb = kp >> 4;
    else if (kp >= F + XE) ntb = FB + ((kp - F - XE) >> 4);
    if (ntb >= 0) {
      float* db = k.bwd + ((int64_t)(stage - 1) * k.S + slot) * (FB * 2 * FB * 256);
      *reinterpret_cast<float4*>(db + ((((col >> 4) * 2 * FB + ntb) * 64) + (kp & 15) + 16 * ((col & 15) >> 2)) * 4) = p;
    }
  }
}

struct AdamArgs {
  AdamPack pack;                                         // pack.fwd == null: no fragment-major copy
  float* param; float* grad; float* mom; float* vel;
  const float* slab; int64_t slab_stride; int n_slabs;   // grad = sum of slabs (if slab != null)
  int n_layers; int64_t layer_end4[16]; int layer_slabs[16];   // slabs written per layer (float4 offsets)
  int64_t n4;                                            // P / 4
  float lr_t, beta1, beta2, eps;
  int do_adam;
  // blocks [n_adam_blocks, gridDim.x) reduce the per-row Huber sums to per-output means
  int n_adam_blocks;
  const float* rowloss; float* loss; int loss_n_idx, loss_stride; float loss_scale;
  int64_t loss_slot_stride;                              // rowloss index of (output slot, i) = slot * loss_slot_stride + i * loss_stride
  int loss_split; float* loss_part; unsigned* loss_cnt;  // ranges per output (> 1 only with a single output)
  int64_t range_begin4, range_end4;                      // float4 range of the flat buffer this launch covers ([0, n4) = all)
  int64_t range2_begin4, range2_end4;                    // optional second range, walked after the first (empty: begin == end)
  const float* grad_direct;                              // where slab-less layers left their gradient
  int groups;                                            // threads per float4 column (1, 4 or 16): split of the slab sum
};

__global__ __launch_bounds__(256) void k_reduce_adam(AdamArgs a) {
  if ((int)blockIdx.x >= a.n_adam_blocks) {      // loss role (deterministic tree reduction)
    __shared__ float red[256];
    __shared__ int is_last;
    const int lb = blockIdx.x - a.n_adam_blocks;
    const int slot = lb / a.loss_split, part = lb - slot * a.loss_split;
    const int per = (a.loss_n_idx + a.loss_split - 1) / a.loss_split;
    const int i1 = min((part + 1) * per, a.loss_n_idx);
    float s = 0.f;
    for (int i = part * per + threadIdx.x; i < i1; i += 256) s += a.rowloss[(int64_t)i * a.loss_stride + (int64_t)slot * a.loss_slot_stride];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (a.loss_split == 1) {
      if (threadIdx.x == 0) a.loss[slot] = red[0] * a.loss_scale;
      return;
    }
    // one long output (ragged batches: a single mean over all node rows) is cut into loss_split ranges; the LAST
    // workgroup to finish adds the partials in range order (agent-scope atomics: partials cross XCD L2s)
    if (threadIdx.x == 0) {
      __hip_atomic_store(a.loss_part + part, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned ticket = __hip_atomic_fetch_add(a.loss_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      is_last = ticket == (unsigned)a.loss_split - 1;
      if (is_last) {
        float t = 0.f;
        for (int k = 0; k < a.loss_split; ++k) t += __hip_atomic_load(a.loss_part + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.loss[slot] = t * a.loss_scale;
        __hip_atomic_store(a.loss_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  // `groups` threads share one float4 column and split its slabs (few parameters + many slabs = shared weights at a
  // large batch: without the split 37 workgroups would each walk ~180 slabs serially); partials are combined through
  // LDS in a fixed order, so the result does not depend on scheduling.
  __shared__ float4 part[256];
  const int G = a.groups, cols = 256 / G;
  const int col = threadIdx.x % cols, grp = threadIdx.x / cols;
  const int64_t len1 = a.range_end4 - a.range_begin4, len_all = len1 + (a.range2_end4 - a.range2_begin4);
  for (int64_t base = (int64_t)blockIdx.x * cols; base < len_all; base += (int64_t)a.n_adam_blocks * cols) {
    const int64_t t = base + col;                         // position in [range 1 | range 2]
    const bool act = t < len_all;
    const int64_t i = t < len1 ? a.range_begin4 + t : a.range2_begin4 + (t - len1);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    bool direct = false;
    if (a.slab) {
      if (act) {
        int l = 0;
        while (l + 1 < a.n_layers && i >= a.layer_end4[l]) ++l;
        const int ns = a.layer_slabs[l];
        // four slabs in flight and four running sums (one load, one dependent add at a time made the column's sum a chain of ns
        // memory round trips); fixed order: (s0 + s1) + (s2 + s3) with s_u = slabs grp + u G, grp + (u + 4) G, ...
        float4 g1 = make_float4(0.f, 0.f, 0.f, 0.f), g2 = g1, g3 = g1;
        int c = grp;
        for (; c + 3 * G < ns; c += 4 * G) {
          const float4 t0 = reinterpret_cast<const float4*>(a.slab + (int64_t)c * a.slab_stride)[i];
          const float4 t1 = reinterpret_cast<const float4*>(a.slab + (int64_t)(c + G) * a.slab_stride)[i];
          const float4 t2 = reinterpret_cast<const float4*>(a.slab + (int64_t)(c + 2 * G) * a.slab_stride)[i];
          const float4 t3 = reinterpret_cast<const float4*>(a.slab + (int64_t)(c + 3 * G) * a.slab_stride)[i];
          g.x += t0.x; g.y += t0.y; g.z += t0.z; g.w += t0.w;
          g1.x += t1.x; g1.y += t1.y; g1.z += t1.z; g1.w += t1.w;
          g2.x += t2.x; g2.y += t2.y; g2.z += t2.z; g2.w += t2.w;
          g3.x += t3.x; g3.y += t3.y; g3.z += t3.z; g3.w += t3.w;
        }
        for (; c < ns; c += G) {
          const float4 t = reinterpret_cast<const float4*>(a.slab + (int64_t)c * a.slab_stride)[i];
          g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
        }
        g.x = (g.x + g1.x) + (g2.x + g3.x); g.y = (g.y + g1.y) + (g2.y + g3.y);
        g.z = (g.z + g1.z) + (g2.z + g3.z); g.w = (g.w + g1.w) + (g2.w + g3.w);
        // a layer with no slabs had its gradient written straight into the gradient buffer (wide path, one row split)
        if (ns == 0 && grp == 0) g = reinterpret_cast<const float4*>(a.grad_direct)[i];
        direct = ns == 0 && a.grad == a.grad_direct;       // already where it belongs: no copy
      }
      if (G > 1) {
        part[threadIdx.x] = g;
        __syncthreads();
        if (grp == 0) {
          for (int k = 1; k < G; ++k) {
            const float4 t = part[k * cols + col];
            g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
          }
        }
        __syncthreads();
      }
      if (act && grp == 0 && !direct) reinterpret_cast<float4*>(a.grad)[i] = g;
    } else if (act && grp == 0) {
      g = reinterpret_cast<const float4*>(a.grad)[i];
    }
    if (a.do_adam && act && grp == 0) {
      float4 m = reinterpret_cast<float4*>(a.mom)[i], v = reinterpret_cast<float4*>(a.vel)[i];
      float4 p = reinterpret_cast<float4*>(a.param)[i];
      const float b1 = a.beta1, b2 = a.beta2, ob1 = 1.f - a.beta1, ob2 = 1.f - a.beta2;
      m.x = b1 * m.x + ob1 * g.x; m.y = b1 * m.y + ob1 * g.y; m.z = b1 * m.z + ob1 * g.z; m.w = b1 * m.w + ob1 * g.w;
      v.x = b2 * v.x + ob2 * (g.x * g.x); v.y = b2 * v.y + ob2 * (g.y * g.y);
      v.z = b2 * v.z + ob2 * (g.z * g.z); v.w = b2 * v.w + ob2 * (g.w * g.w);
      p.x -= a.lr_t * m.x / (sqrtf(v.x) + a.eps); p.y -= a.lr_t * m.y / (sqrtf(v.y) + a.eps);
      p.z -= a.lr_t * m.z / (sqrtf(v.z) + a.eps); p.w -= a.lr_t * m.w / (sqrtf(v.w) + a.eps);
      reinterpret_cast<float4*>(a.mom)[i] = m;
      reinterpret_cast<float4*>(a.vel)[i] = v;
      reinterpret_cast<float4*>(a.param)[i] = p;
      if (a.pack.fwd) pack_scatter(a.pack, 4 * i, p);
    }
  }
}

// dst[slot * slot_stride + i] = 0 for i < count (count, slot_stride multiples of 4): zero gradient of weight rows
// whose input segment is absent
__global__ __launch_bounds__(256) void k_zero_rows(float* dst, int64_t slot_stride, int64_t count) {
  float* p = dst + blockIdx.y * slot_stride;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += (int64_t)gridDim.x * 1024)
    *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// scalar tail-safe Adam for arbitrary n (v2x_adam_step entry point)
__global__ __launch_bounds__(256) void k_adam_scalar(float* p, const float* g, float* m, float* v,
                                                     int64_t n, float lr_t, float b1, float b2, float eps) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * (gi * gi);
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

// per-output Huber mean: loss[k] = sum_b rowloss[b*N + k]   (already scaled by 1/(B*C) here)
__global__ __launch_bounds__(256) void k_loss_reduce(const float* rowloss, float* loss, int n_idx,
                                                     int row_stride, int64_t slot_stride, float inv_denom) {
  __shared__ float red[256];
  const int slot = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n_idx; i += 256) s += rowloss[(int64_t)i * row_stride + slot * slot_stride];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[slot] = red[0] * inv_denom;
}

// =====================================================================================
// DQN replay glue (Agent.replay, BS_brain.py:573-692) for transitions resident in HBM
// =====================================================================================
// dst[i] = src[idx[i]], rows of `words` 4-byte words; VEC = 4 when rows are 16-byte multiples (coalesced float4)
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_rows(const uint32_t* src, const int32_t* idx, uint32_t* dst,
                                                     int64_t n_idx, int64_t words) {
  const int64_t per_row = words / VEC;
  const int64_t total = n_idx * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per_row, c = (i - r * per_row) * VEC;
    const int64_t s = (int64_t)idx[r] * words + c;
    if (VEC == 4) *reinterpret_cast<uint4*>(dst + r * words + c) = *reinterpret_cast<const uint4*>(src + s);
    else dst[r * words + c] = src[s];
  }
}

// Several gathers by the SAME index list in one launch (the replay minibatch: s, s', action, reward, CSR sources --
// Agent.replay's per-sample loops, BS_brain.py:573-640): blockIdx.y selects the job.
constexpr int GATHER_MAX_JOBS = 8;
struct GatherJobs { const uint32_t* src[GATHER_MAX_JOBS]; uint32_t* dst[GATHER_MAX_JOBS]; int64_t words[GATHER_MAX_JOBS]; int vec[GATHER_MAX_JOBS]; };
__global__ __launch_bounds__(256) void k_gather_rows_multi(GatherJobs jobs, const int32_t* idx, int64_t n_idx) {
  typedef const __attribute__((address_space(4))) uint64_t* CQ;
  CQ kq = (CQ)__builtin_amdgcn_kernarg_segment_ptr();            // (runtime-indexed kernarg arrays would be copied to scratch)
  const int jb = blockIdx.y;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(kq[offsetof(GatherJobs, src) / 8 + jb]);
  uint32_t* dst = reinterpret_cast<uint32_t*>(kq[offsetof(GatherJobs, dst) / 8 + jb]);
  const int64_t words = (int64_t)kq[offsetof(GatherJobs, words) / 8 + jb];
  const int vec = ((const __attribute__((address_space(4))) int*)kq)[offsetof(GatherJobs, vec) / 4 + jb];
  const int64_t per_row = vec ? words / 4 : words;
  const int64_t total = n_idx * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per_row, c = (i - r * per_row) * (vec ? 4 : 1);
    const int64_t s = (int64_t)idx[r] * words + c;
    if (vec) *reinterpret_cast<uint4*>(dst + r * words + c) = *reinterpret_cast<const uint4*>(src + s);
    else dst[r * words + c] = src[s];
  }
}

// Q statistics of a minibatch of fitted targets y[B][N][C] (BS_brain.py:743-746: per link the mean of all entries and the mean of
// the per-sample maxima), as float64 SUMS per link: out[0][k] = sum_b sum_c y, out[1][k] = sum_b max_c y.  QS_PARTS workgroups
// per link, each a fixed range of the samples (every thread a fixed subset, combined through LDS in a fixed order); the last
// one to arrive adds the parts in index order: deterministic.  (One workgroup per link was 17 us at 4096 samples x 20 links: 20
// workgroups walking rows 320 bytes apart.)  part: [N][QS_PARTS][2] doubles, cnt: [N] ints, zero before the first launch.
constexpr int QS_PARTS = 8;
__global__ __launch_bounds__(256) void k_q_stats(const float* y, int B, int N, int C, double* out, double* part, unsigned* cnt) {
  __shared__ double s_all[256], s_max[256];
  unsigned s_last;
  const int k = blockIdx.x, pz = blockIdx.y;
  const int b0 = (int)((int64_t)B * pz / QS_PARTS), b1 = (int)((int64_t)B * (pz + 1) / QS_PARTS);
  double a = 0.0, m = 0.0;
  for (int b = b0 + threadIdx.x; b < b1; b += 256) {
    const float* row = y + ((int64_t)b * N + k) * C;
    float mx = row[0];
    double s = (double)row[0];
    for (int c = 1; c < C; ++c) { s += (double)row[c]; mx = fmaxf(mx, row[c]); }
    a += s;
    m += (double)mx;
  }
  s_all[threadIdx.x] = a; s_max[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { s_all[threadIdx.x] += s_all[threadIdx.x + st]; s_max[threadIdx.x] += s_max[threadIdx.x + st]; }
    __syncthreads();
  }
  // (agent-scope atomics, as in k_reduce_adam's loss role: the parts cross XCD L2s)
  if (threadIdx.x == 0) {
    __hip_atomic_store(part + ((int64_t)k * QS_PARTS + pz) * 2, s_all[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + ((int64_t)k * QS_PARTS + pz) * 2 + 1, s_max[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned ticket = __hip_atomic_fetch_add(cnt + k, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == QS_PARTS - 1 ? 1u : 0u;
    if (s_last) {
      double ta = 0.0, tm = 0.0;
      for (int q = 0; q < QS_PARTS; ++q) {
        ta += __hip_atomic_load(part + ((int64_t)k * QS_PARTS + q) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tm += __hip_atomic_load(part + ((int64_t)k * QS_PARTS + q) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      out[k] = ta; out[N + k] = tm;
      __hip_atomic_store(cnt + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
    }
  }
}

// the replaced entry of the target rule and the action, one float4 per row (MlpArgs::tq): tq[row] = {r[graph] + gamma * max_c
// q_next[row][c] (evaluated like k_dqn_targets below), action[row] as bits, 0, 0}
__global__ __launch_bounds__(256) void k_dqn_tq(const float* qn, const int32_t* action, const double* reward, double gamma, int n_rows,
                                                int n_nodes, int C, float* tq) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  float mx = qn[(int64_t)row * C];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, qn[(int64_t)row * C + c]);
  const float t = (float)(reward[row / n_nodes] + gamma * (double)mx);
  *reinterpret_cast<float4*>(tq + (int64_t)row * 4) = make_float4(t, __int_as_float(action[row]), 0.f, 0.f);
}

// one thread per (graph, node) row
__global__ __launch_bounds__(256) void k_dqn_targets(const float* q, const float* qn, const int32_t* action,
                                                     const double* reward, double gamma, int n_rows, int n_nodes,
                                                     int C, float* y) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  float mx = qn[(int64_t)row * C];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, qn[(int64_t)row * C + c]);
  // `r + GAMMA * np.amax(p_)` (BS_brain.py:690) scalar by scalar with p_ float32: under the reference's numpy-1.x
  // stack the product is float64 (python float x np.float32), rounded to fp32 once when stored.  (numpy >= 2 / NEP 50
  // rounds the product to fp32 first; the fixtures captured under numpy 2.2.6 differ from this by <= 1 ulp.)
  const float t = (float)(reward[row / n_nodes] + gamma * (double)mx);
  const int a = action[row];
  for (int c = 0; c < C; ++c) y[(int64_t)row * C + c] = c == a ? t : q[(int64_t)row * C + c];
}

// Contract check of a device-resident batch (include/v2xgnn.h "Data layout"): every graph has 1..max_nodes rows and at
// most max_edges edges, row_ptr is monotone, every source id lies inside its graph and the sources of a row are strictly
// ascending (=> no duplicate edges: the backward pass de-duplicates through bit masks, the forward pass would not).
// One workgroup per graph (grid-stride); flag bits: 1 sizes, 2 row_ptr, 4 source range, 8 order / duplicates.
__global__ __launch_bounds__(256) void k_validate_batch(const int32_t* graph_off, const int32_t* row_ptr, const int32_t* col_idx,
                                                        int n_graphs, int n_rows, int n_edges, int n_nodes, int max_nodes,
                                                        int max_edges, int* flag) {
  int bad = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (row_ptr[0] != 0 || row_ptr[n_rows] != n_edges) bad |= 2;
    if (graph_off && (graph_off[0] != 0 || graph_off[n_graphs] != n_rows)) bad |= 1;
  }
  for (int g = blockIdx.x; g < n_graphs; g += gridDim.x) {
    const int r0 = graph_off ? graph_off[g] : g * n_nodes;
    const int r1 = graph_off ? graph_off[g + 1] : r0 + n_nodes;
    const int n = r1 - r0;
    if (n < 1 || n > max_nodes || r0 < 0 || r1 > n_rows) { bad |= 1; continue; }
    const int e0 = row_ptr[r0], e1 = row_ptr[r1];
    if (e1 < e0 || e1 - e0 > max_edges || e0 < 0 || e1 > n_edges) { bad |= 2; continue; }
    for (int q = r0 + threadIdx.x; q < r1; q += 256) {
      const int a0 = row_ptr[q], a1 = row_ptr[q + 1];
      if (a1 < a0 || a0 < e0 || a1 > e1) { bad |= 2; continue; }
      int prev = -1;
      for (int e = a0; e < a1; ++e) {
        const int c = col_idx[e];
        if (c < 0 || c >= n) bad |= 4;
        if (c <= prev) bad |= 8;
        prev = c;
      }
    }
  }
  if (bad) atomicOr(flag, bad << 4);
}

}  // namespace v2x
