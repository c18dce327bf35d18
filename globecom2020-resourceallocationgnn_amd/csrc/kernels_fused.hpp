// Graph-major FUSED graph layers for fixed-size graphs (the headline configuration: 20 links, per-node weights).
//
// kernels.hpp runs a layer as two launches per direction -- aggregation (graph-major, AggLayer.call
// BS_brain.py:69-76) and node update (slot-major, GNNLayer.call :44-51) -- because with per-node weights a workgroup
// wants ONE slot's weight image in LDS.  Here a workgroup owns 16 whole graphs instead, so that the 16 rows of node
// slot k (one per graph) are exactly one 16-row MFMA tile:
//   * the tile's feature rows h_s live in LDS for the whole layer (the gather of AggLayer never leaves the CU),
//   * slot k's weights are NOT staged in LDS: they stream from L2 straight into MFMA A fragments, from a
//     "fragment-major" copy of the parameters (k_pack_weights) in which every load is one coalesced 1 KiB wave access
//     (tools/l2stream.hip: 25-33 TB/s of L2 reads are available, 8 B of weights per MFMA flop need 19 TB/s at peak),
//   * all L+1 stages and L+1 aggregations of the forward pass are ONE launch (was 6 at L = 2), the L+1 transposed
//     aggregations and L data gradients of the backward pass ONE launch (was 5); the activations h_s / a_s and the
//     pre-activation gradients dpre_s are still written once each, because the weight-gradient launch (slot-major,
//     k_wgrad) and the decision MLP read them.
// 8 waves per workgroup = 2 per SIMD: while one wave issues its weight loads and gathers from LDS, the other one's
// MFMAs keep the matrix pipe busy (a wave issues in order: tools/l2stream.hip measured 0.55-0.7 of the MFMA peak for a
// single wave that prefetches for itself, and clocks that do NOT drop when both pipes run).
// Arithmetic order is that of k_agg_small / k_gemm_rows, so the results are bitwise those of the unfused path.
#pragma once
#include "kernels.hpp"

namespace v2x {

constexpr int FZ_TG = 16;        // graphs per workgroup = rows of one slot's MFMA tile
constexpr int FZ_WAVES = 8;
constexpr int FZ_THREADS = 64 * FZ_WAVES;
constexpr int FZ_MAXL = 8;

typedef const __attribute__((address_space(1))) f32x4* gvec_p;

// sizes of the fragment-major parameter copy (floats)
template <int F>
struct FzPack {
  static constexpr int FB = F / 16, KB = 2 * FB + 1;
  static constexpr int FWD0 = FB * 256 + F;             // stage 0: [1 k-block][FB n-tiles][64 lanes][4] + bias[F]
  static constexpr int FWD = KB * FB * 256 + F;         // stage >= 1: [KB][FB][64][4] + bias[F]
  static constexpr int BWD = FB * 2 * FB * 256;         // stage >= 1: [FB k-blocks][2 FB n-tiles][64][4]
  static constexpr int ROWF = 4 * (FB | 1);             // LDS row of one k-group: FB float4 + pad to an ODD float4 count
};

struct PackArgs {
  const float* params; float* pk_fwd; float* pk_bwd;
  int64_t layer_off[FZ_MAXL + 1]; int64_t slot_stride[FZ_MAXL + 1]; RowPad pad[FZ_MAXL + 1];
  int S;
};

// forward fragment  (stage, slot, kb, nt, lane, s) = W[real_row(kb*16 + 4*(lane>>4) + s)][nt*16 + (lane&15)]
// backward fragment (stage, slot, kb, nt, lane, s) = W[real_row(orow(nt) + (lane&15))][kb*16 + 4*(lane>>4) + s],
//   orow = the h rows (nt < FB) and the agg rows (nt >= FB) of the layer's padded K axis.
// grid = (x, S, L+1)
template <int F>
__global__ __launch_bounds__(256) void k_pack_weights(PackArgs a) {
  using P = FzPack<F>;
  constexpr int FB = P::FB;
  typedef const __attribute__((address_space(4))) unsigned char* CBytes;
  CBytes kp = (CBytes)__builtin_amdgcn_kernarg_segment_ptr();
  const int stage = blockIdx.z, slot = blockIdx.y, S = a.S;
  const int64_t loff = *(const __attribute__((address_space(4))) int64_t*)(kp + offsetof(PackArgs, layer_off) + 8 * stage);
  const int64_t sstr = *(const __attribute__((address_space(4))) int64_t*)(kp + offsetof(PackArgs, slot_stride) + 8 * stage);
  RowPad pad;
  pad.pad_at = *(const __attribute__((address_space(4))) int*)(kp + offsetof(PackArgs, pad) + 12 * stage);
  pad.n_pad = *(const __attribute__((address_space(4))) int*)(kp + offsetof(PackArgs, pad) + 12 * stage + 4);
  pad.k_real = *(const __attribute__((address_space(4))) int*)(kp + offsetof(PackArgs, pad) + 12 * stage + 8);
  const float* W = a.params + loff + slot * sstr;
  const int kbs = stage ? P::KB : 1;
  float* dst = a.pk_fwd + (stage ? (int64_t)S * P::FWD0 + ((int64_t)(stage - 1) * S + slot) * P::FWD : (int64_t)slot * P::FWD0);
  const int n_f = kbs * FB * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_f + F; i += gridDim.x * 256) {
    float v;
    if (i < n_f) {
      const int s = i & 3, lane = (i >> 2) & 63, c = i >> 8, nt = c % FB, kb = c / FB;
      const int rr = real_row(pad, kb * 16 + 4 * (lane >> 4) + s);
      v = rr >= 0 ? W[(int64_t)rr * F + nt * 16 + (lane & 15)] : 0.f;
    } else {
      v = W[(int64_t)pad.k_real * F + (i - n_f)];
    }
    dst[i] = v;
  }
  if (stage == 0) return;
  float* dstb = a.pk_bwd + ((int64_t)(stage - 1) * S + slot) * P::BWD;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P::BWD; i += gridDim.x * 256) {
    const int s = i & 3, lane = (i >> 2) & 63, c = i >> 8, nt = c % (2 * FB), kb = c / (2 * FB);
    const int orow = nt < FB ? nt * 16 : F + XE + (nt - FB) * 16;
    const int rr = real_row(pad, orow + (lane & 15));
    dstb[i] = W[(int64_t)rr * F + kb * 16 + 4 * (lane >> 4) + s];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS tile of the 16 graphs' feature rows, split by MFMA k-group so that every ds_read_b128 / ds_write_b128 of a
// wave is conflict-free:  element (node p, graph j, feature f) lives at
//     sub-array kg = (f % 16) / 4,  row p*16 + j,  float4 slot f / 16        (ROWF floats per row, an odd float4 count)
// A 16-lane access group of a b128 LDS instruction holds 16 distinct graphs j (k-groups 0/1 or 2/3 mixed), the
// sub-array stride and 16*ROWF are multiples of 64 banks, so the bank group is (ROWF4 * j + const) mod 16: distinct
// for distinct j whatever the nodes p_j the lanes gather from.
// ---------------------------------------------------------------------------------------------------------------
struct FusedFwdArgs {
  const float* xe; const int32_t* row_ptr; const int32_t* col_idx;
  const float* pk;                                   // fragment-major forward weights (k_pack_weights)
  float* h[FZ_MAXL + 1]; float* a[FZ_MAXL + 1];      // stage outputs h_s and their aggregations a_s, [R][F]
  unsigned short* gate[FZ_MAXL + 1];                 // sign bits of h_s (ReLU' gates, s < L) for the fused backward: 16 bits per lane,
                                                     // [node][workgroup][64 lanes] (bit 4 nt + c = column nt*16 + 4 kg + c > 0)
  int n_graphs, N, L, S;                             // S = weight slots (N per-node, 1 shared)
  int edges_cap;                                     // LDS bytes reserved for the tile's edge list (16 * max_edges)
  int n_edges;                                       // E of the whole batch
  int compl_sums;                                    // (informative; the COMPL kernel instance is what runs) dense graphs: Agg(h)[q] = colsum(h) - sum over the NON-neighbours of q
  unsigned* nbmask;                                  // the rows' in-neighbour sets (complement form: NON-neighbour sets), [R] words, for the fused backward
  int frag_out;                                      // h_L and a_L fragment-major for k_mlp_train_wg (MlpArgs::frag_groups); L >= 1
  int* err;
  long long* ts;                                     // TS builds: [8 waves][64] 100 MHz time stamps of workgroup 7
};

// Aggregation through the complement (compl_sums).  The reference's interference graph is almost complete: link q hears
// every other link but itself and one more (in-degree N - 2), so the gather of 18 neighbour rows per node -- LDS-bound,
// 5 us per aggregation, 6 aggregations per fit step -- is replaced by   Agg[q] = S - sum_{p not in N(q)} row[p]   with
// S the column sum of the graph's rows: two rows instead of eighteen.  S costs nothing to produce: every wave adds up
// the output rows of ITS slots while they are still in registers and parks that partial sum ([wave][kg][graph] float4
// rows, same conflict-free geometry as the tile) next to the rows it writes into the tile; a lane then adds the 8
// partials in wave order (fixed order: deterministic).  The host selects this form when the average in-degree exceeds
// (N - 1) / 2; the values differ from the edge-ordered gather by rounding only.
// The 8 partials are added ONCE per stage, in wave order (fixed order: deterministic), by all threads together (a
// float2 each) between the two barriers that replace the tile, into a 64-row table of totals that the gathers read: a
// lane that adds the 8 partials itself reads 32 float4 per sum -- 8x redundant over the workgroup, and the gather phases
// are LDS-bandwidth-bound (the backward re-added them per slot: 0.8 MB of LDS reads per workgroup and stage).
constexpr int FZ_SUMS_ROWS = FZ_WAVES * 4 * FZ_TG;     // rows of ROWF floats: the partials
constexpr int FZ_TOT_ROWS = 4 * FZ_TG;                 // the totals

template <int FB, int ROWF>
__device__ __forceinline__ void fz_reduce_sums(const float* sS, float* sT) {
  for (int e = threadIdx.x; e < FZ_TOT_ROWS * FB * 2; e += FZ_THREADS) {
    const int off = (e / (2 * FB)) * ROWF + 2 * (e % (2 * FB));
    float2 v = *reinterpret_cast<const float2*>(sS + off);
#pragma unroll
    for (int w = 1; w < FZ_WAVES; ++w) {
      const float2 t = *reinterpret_cast<const float2*>(sS + w * FZ_TOT_ROWS * ROWF + off);
      v.x += t.x; v.y += t.y;
    }
    *reinterpret_cast<float2*>(sT + off) = v;
  }
}

// phase time stamps of one workgroup (measurement builds only; v2x_debug_phase_stamps)
template <bool TS>
struct FzStamp {
  long long* p; int n;
  __device__ __forceinline__ FzStamp(long long* base, int wv, int lane) : p(nullptr), n(0) {
    if (TS && base && blockIdx.x == 7 && lane == 0) p = base + wv * 64;
  }
  __device__ __forceinline__ void mark(bool drain = false) {
    if (TS) {
      if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (p && n < 64) p[n] = wall_clock64();
      ++n;
    }
  }
};

// explicit GLOBAL accesses: pointers rebuilt from the kernarg segment are generic, and a FLAT store / load makes the
// wait-count pass give up counting (vmcnt(0) in front of every dependent MFMA)
typedef __attribute__((address_space(1))) f32x4* gvec_wp;
// The rows these kernels store (h_s, a_s, dpre_s: 126 + 63 MB per step at the headline size) are read again one to three
// launches later by the weight-gradient launch -- far more than the 32 MB of L2 holds -- so they are stored NON-TEMPORAL
// (`global_store ... nt`): they do not displace the weight fragments and partial-sum slabs that ARE re-read from L2.
// Measured A/B on one box (round 3): step 0.2541 -> 0.2502 ms (the slab sums + Adam 15.5 -> 13.8 us, backward - 0.8 us,
// forward + 0.8 us); non-temporal LOADS of the same rows in k_wgrad: no gain (0.2505 vs 0.2508), not used.
#ifndef V2X_NT_STORES
#define V2X_NT_STORES 1
#endif
#if V2X_NT_STORES
__device__ __forceinline__ void stg4(float* p, f32x4 v) { __builtin_nontemporal_store(v, (gvec_wp)p); }
#else
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *(gvec_wp)p = v; }
#endif
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(gvec_p)p; }

// ReLU' gates as bits: the backward needs h_s only to know where it is positive.  Reading the rows back costs it 21 MB per
// stage, requested by 256 lock-stepped workgroups at the moment they are needed (measured with the gate removed: 7 us of
// the backward's 51); 16 bits per lane are 0.65 MB per stage and arrive a whole stage ahead.
__device__ __forceinline__ unsigned gate_bits4(f32x4 v) {
  return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}
__device__ __forceinline__ f32x4 gate_apply4(f32x4 g, unsigned m) {
  return (f32x4){(m & 1u) ? g[0] : 0.f, (m & 2u) ? g[1] : 0.f, (m & 4u) ? g[2] : 0.f, (m & 8u) ? g[3] : 0.f};
}

// Loads and stores share ONE in-order counter (vmcnt) but are acknowledged independently: with a store in flight a load
// cannot be waited for by count, the wait becomes vmcnt(0) -- for the store's acknowledgement AND for every load
// requested since, i.e. the weight ring loses its prefetch distance.  Hence: no store inside a run of MFMAs that waits
// for weight chunks (they go in front of it, in the gather phase, or behind it), and the first MFMA of a run is issued
// before the run's first new request.  The barriers of these kernels order LDS accesses only (__syncthreads() would also
// drain vmcnt, i.e. wait for the stores just issued): nothing global is exchanged between the waves.
__device__ __forceinline__ void fz_barrier() {
#ifdef V2X_FZ_FULL_BARRIER
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

__device__ __forceinline__ f32x4 ldnt4(const float* p) {      // L1-bypassing load (data another wave of this CU just wrote)
  return __builtin_nontemporal_load((gvec_p)p);
}

// Per stage a wave (1) gathers the neighbour sums of ALL its slots from LDS (all 8 waves do this at the same time: the
// phase is LDS-bound and no MFMA waits behind it), (2) runs its slots' MFMAs back to back while the NEXT slot's weight
// fragments stream in through a 3-chunk register ring (one 1 KiB load per 4 MFMAs: a wave issues in order, a burst of
// loads in front of the MFMAs idles the matrix pipe -- tools/l2stream.hip), keeping the outputs in registers, and
// (3) after a barrier writes them into the LDS tile for the next stage.  No global -> LDS reload between stages.
//
// NS = number of slots of THIS wave, a compile-time constant: the per-slot register arrays need constant indices, and
// the MFMA phase must be straight-line code (a wave-uniform `if (slot exists)` between two slots makes the compiler's
// wait-count pass merge two load histories and fall back to vmcnt(0) in front of the MFMAs, which stalls the ring).
// The kernel instantiates the body for ceil(N/8) and ceil(N/8) - 1 slots and every wave picks one, once.  For the same
// reason nothing is predicated per lane: lanes of graphs past the end of the batch shadow the last real graph (same
// addresses, same values), so their stores are harmless duplicates.
// CSR slice of the tile -> LDS: edge offsets relative to the tile, sources as bytes.  The loads are on the critical path
// of the launch (kernel start -> row_ptr[first row] -> col_idx slice -> LDS -> first gather), so they are issued before
// anything else and do not wait for row_ptr: the slice is fetched from the offset it has when every graph holds max_edges
// edges (always true for the reference topology) and only re-fetched if that guess turns out wrong.
constexpr int FZ_CSR_U = 12;      // early source loads per thread: 512 x 12 = 6144 edges (16 graphs x 360 = 5760)
struct FzCsrEarly { int cv[FZ_CSR_U]; int rpv[2]; int e_guess; };
__device__ __forceinline__ void csr_issue(FzCsrEarly& c, const int32_t* row_ptr, const int32_t* col_idx, int r_begin, int nrows,
                                          int g0, int edges_cap, int n_edges_total) {
  c.e_guess = g0 * (edges_cap / FZ_TG);
#pragma unroll
  for (int u = 0; u < FZ_CSR_U; ++u) c.cv[u] = col_idx[max(min(c.e_guess + (int)threadIdx.x + u * FZ_THREADS, n_edges_total - 1), 0)];
#pragma unroll
  for (int u = 0; u < 2; ++u) c.rpv[u] = row_ptr[r_begin + min((int)threadIdx.x + u * FZ_THREADS, nrows)];
}
__device__ __forceinline__ void csr_commit(const FzCsrEarly& c, const int32_t* row_ptr, const int32_t* col_idx, int* sRp,
                                           unsigned char* sCol, int N, int r_begin, int nrows, int e_begin, int nedges) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = threadIdx.x + u * FZ_THREADS;
    if (i <= FZ_TG * N) sRp[i] = i <= nrows ? c.rpv[u] - e_begin : nedges;
  }
  for (int i = threadIdx.x + 2 * FZ_THREADS; i <= FZ_TG * N; i += FZ_THREADS) sRp[i] = i <= nrows ? row_ptr[r_begin + i] - e_begin : nedges;
  int done = 0;
  if (e_begin == c.e_guess) {                                    // workgroup-uniform
#pragma unroll
    for (int u = 0; u < FZ_CSR_U; ++u) {
      const int i = threadIdx.x + u * FZ_THREADS;
      if (i < nedges) sCol[i] = (unsigned char)min((unsigned)c.cv[u], (unsigned)(N - 1));
    }
    done = FZ_CSR_U * FZ_THREADS;
  }
  for (int base = done + threadIdx.x; base < nedges; base += FZ_THREADS * 6) {
    int v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = col_idx[e_begin + min(base + u * FZ_THREADS, nedges - 1)];
#pragma unroll
    for (int u = 0; u < 6; ++u)
      if (base + u * FZ_THREADS < nedges) sCol[base + u * FZ_THREADS] = (unsigned char)min((unsigned)v[u], (unsigned)(N - 1));
  }
}

struct FzCtx {
  float* sH; int* sRp; unsigned char* sCol; float* sS; float* sT; unsigned* sC;
  int N, L, SUB, lane, wv, jc, kg, g0;
};

template <int F, int NS, bool TS, bool COMPL>
__device__ __forceinline__ void fused_fwd_body(const FusedFwdArgs& a, const FzCtx& x, const int nrows, const int r_begin) {
  using P = FzPack<F>;
  constexpr int FB = P::FB, KB = P::KB, ROWF = P::ROWF;
  constexpr bool RING = KB % 3 == 0 && NS > 0;                   // F = 64: 9 k-blocks = 3 chunks of 3; F = 16: 3 chunks of 1
  constexpr int NCH = RING ? 3 : 1, CKB = KB / NCH, CHN = CKB * FB;   // chunks per slot, k-blocks / float4 per chunk
  constexpr int NSA = NS > 0 ? NS : 1;
  const int N = x.N, L = x.L, lane = x.lane, wv = x.wv, kg = x.kg, jc = x.jc;
  FzStamp<TS> ts(a.ts, wv, lane);
  ts.mark();                                                     // 0: start

  int64_t rowi[NSA];
#pragma unroll
  for (int i = 0; i < NS; ++i) rowi[i] = (int64_t)(x.g0 + jc) * N + (wv + FZ_WAVES * i);
  // fragment-major weights of (stage s >= 1, this wave's i-th slot); the item after the last one aliases a real one
  auto item_base = [&](int s, int i) -> const float* {
    return a.pk + (int64_t)a.S * P::FWD0 + ((int64_t)(min(s, L) - 1) * a.S + (a.S == 1 ? 0 : wv + FZ_WAVES * i)) * P::FWD;
  };
  f32x4 wr[NCH][CHN];
  auto wload = [&](int c, const float* base) {
    gvec_p wp = (gvec_p)base + lane + c * CHN * 64;
#pragma unroll
    for (int u = 0; u < CHN; ++u) wr[c][u] = wp[u * 64];
  };
  // ---- requests in the order of their urgency: CSR slice, embed operands, the first weight chunks of stage 1
  FzCsrEarly csr;
  csr_issue(csr, a.row_ptr, a.col_idx, r_begin, nrows, x.g0, a.edges_cap, a.n_edges);
  f32x4 xev[NSA], w0[NSA][FB], b0[NSA][FB];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const float* wb = a.pk + (int64_t)(a.S == 1 ? 0 : wv + FZ_WAVES * i) * P::FWD0;
    xev[i] = ldg4(a.xe + rowi[i] * XE + 4 * kg);
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) { w0[i][nt] = ((gvec_p)wb + lane)[nt * 64]; b0[i][nt] = ldg4(wb + FB * 256 + nt * 16 + 4 * kg); }
  }
  if (RING && L >= 1) { wload(0, item_base(1, 0)); wload(1, item_base(1, 0)); }
  const int e_begin = a.row_ptr[r_begin], nedges = a.row_ptr[r_begin + nrows] - e_begin;
  if (nedges > a.edges_cap || nedges < 0) {                      // workgroup-uniform, before any barrier
    if (threadIdx.x == 0 && a.err) atomicOr(a.err, 1);
    return;
  }
  ts.mark();                                                     // (TS) row_ptr[first], row_ptr[last] known
  csr_commit(csr, a.row_ptr, a.col_idx, x.sRp, x.sCol, N, r_begin, nrows, e_begin, nedges);
  ts.mark();                                                     // (TS) CSR slice in LDS
  constexpr bool compl_sums = COMPL;     // (a compile-time form: both gathers side by side cost registers and spill)
  {                                      // the CSR rows of the tile as bit sets (while the embed operands are in flight): the
    __syncthreads();                     // in-neighbours of a row (edge form) / its NON-neighbours (complement form)
    const unsigned valid = N >= 32 ? 0xffffffffu : (1u << N) - 1u;
    for (int r = threadIdx.x; r < FZ_TG * N; r += FZ_THREADS) {
      unsigned nb = 0u;
      if (r < nrows)
        for (int e = x.sRp[r]; e < x.sRp[r + 1]; ++e) nb |= 1u << x.sCol[e];
      x.sC[r] = compl_sums ? (~nb & valid) : nb;
      if (a.nbmask && r < nrows) a.nbmask[r_begin + r] = compl_sums ? (~nb & valid) : nb;    // for the backward: no CSR there
    }
    ts.mark();                                                   // (TS) masks built
  }

  typedef const __attribute__((address_space(4))) uint64_t* CQ;
  CQ kq = (CQ)__builtin_amdgcn_kernarg_segment_ptr();
  auto hptr = [&](int s) { return reinterpret_cast<float*>(kq[offsetof(FusedFwdArgs, h) / 8 + s]); };
  auto aptr = [&](int s) { return reinterpret_cast<float*>(kq[offsetof(FusedFwdArgs, a) / 8 + s]); };
  auto gptr = [&](int s) { return reinterpret_cast<unsigned short*>(kq[offsetof(FusedFwdArgs, gate) / 8 + s]); };
  const int gate_lane = blockIdx.x * 64 + lane, gate_stride = gridDim.x * 64;     // + node * gate_stride
  float* myrow = x.sH + kg * x.SUB + jc * ROWF;                  // + p*16*ROWF + kb*4
  float* mysum = x.sS + ((wv * 4 + kg) * FZ_TG + jc) * ROWF;     // this wave's partial column sum (compl_sums)
  const float* tot = x.sT + (kg * FZ_TG + jc) * ROWF;            // column sums of the tile (compl_sums)

  // ---- stage 0 (embed): h_0 = relu(xe . W0 + b0); the neighbour-init block is absent (always zero in the reference)
  f32x4 psum[FB];                        // sum of this wave's output rows of the current stage
#pragma unroll
  for (int nt = 0; nt < FB; ++nt) psum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    float* hp = hptr(0);
    unsigned short* gp = gptr(0);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int k = wv + FZ_WAVES * i;
      unsigned gb = 0u;
      f32x4 acc[FB];
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) acc[nt] = V2X_MFMA(w0[i][nt][s], xev[i][s], acc[nt]);
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) {
        const f32x4 v = relu4(acc[nt] + b0[i][nt]);
        stg4(hp + rowi[i] * F + nt * 16 + 4 * kg, v);
        st4(myrow + k * FZ_TG * ROWF + nt * 4, v);
        psum[nt] = i == 0 ? v : psum[nt] + v;
        gb |= gate_bits4(v) << (4 * nt);
      }
      gp[k * gate_stride + gate_lane] = (unsigned short)gb;
    }
    if (compl_sums) {
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(mysum + nt * 4, NS > 0 ? psum[nt] : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
  }
  ts.mark();                                                     // 1: embed done (before barrier)
  __syncthreads();
  if (compl_sums) { fz_reduce_sums<FB, ROWF>(x.sS, x.sT); __syncthreads(); }
  ts.mark();                                                     // 2: after barrier

  // Edge form of AggLayer.call (BS_brain.py:69-76) for ALL of this wave's slots of the lane's graph at once:
  //     a[i][kb] = sum over the in-edges p -> k_i of row p,   ascending sources (k_agg_small's order: bitwise its sums).
  // Round 4.  Until now every slot walked its own CSR row and read its ~N - 2 source rows from LDS -- 18 x 4 b128 reads per
  // slot, and a wave's three slots (same graph!) read nearly the same 20 rows three times: the phase was LDS-bandwidth
  // bound (4.6-5.3 us against 1.8 us for the complement form).  With N <= 32 a CSR row is a 32-bit SET: the wave walks the
  // graph's rows p = 0 .. N - 1 once, reads row p once, and adds it to the slots whose set holds p (multiply by the 0 / 1
  // bit: x * 1 + acc and x * 0 + acc are exact for finite x -- the backward's tail has always done the same): 20 x 4
  // reads instead of 54 x 4, same values, same order.  Cost independent of the degree (<= N rows): sparse graphs read at
  // most what a complete graph reads.
  auto gather_all = [&](f32x4 (&ag)[NSA][FB]) {
    unsigned msk[NSA];
#pragma unroll
    for (int i = 0; i < NSA; ++i) {
      msk[i] = i < NS ? x.sC[jc * N + wv + FZ_WAVES * i] : 0u;
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) ag[i][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // Degree-aware (round 5): a lane whose slots' sets together hold fewer than N / 2 sources walks only those (ascending:
    // the same sums in the same order -- skipping "+ x * 0" is exact, a running sum that started at +0 is never -0); the wave
    // takes the plain walk of all N rows when no lane is that sparse (the reference's complete-minus-two graphs: unchanged
    // code on the headline path).  A 28-link graph of in-degree 2 then costs ~6 row reads per wave and graph, not 28.
    unsigned orm = 0u;
#pragma unroll
    for (int i = 0; i < NS; ++i) orm |= msk[i];
    if (__any(2 * __popc(orm) < N)) {
      unsigned walk = 2 * __popc(orm) < N ? orm : (N >= 32 ? 0xffffffffu : (1u << N) - 1u);
      while (walk) {
        const int p = __builtin_ctz(walk);
        walk &= walk - 1;
        const float* bp = myrow + p * (FZ_TG * ROWF);
        f32x4 v[FB];
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) v[kb] = ld4(bp + kb * 4);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const float f = (float)((msk[i] >> p) & 1u);
#pragma unroll
          for (int kb = 0; kb < FB; ++kb) ag[i][kb] += v[kb] * f;
        }
      }
      return;
    }
    for (int p = 0; p < N; ++p) {
      const float* bp = myrow + p * (FZ_TG * ROWF);
      f32x4 v[FB];
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) v[kb] = ld4(bp + kb * 4);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const float f = (float)((msk[i] >> p) & 1u);
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) ag[i][kb] += v[kb] * f;
      }
    }
  };

  // the same through the complement: column sum minus the rows of the non-neighbours
  f32x4 csum[FB];
  auto colsum = [&]() {
#pragma unroll
    for (int kb = 0; kb < FB; ++kb) csum[kb] = ld4(tot + kb * 4);
  };
  auto gather_c = [&](int k, f32x4 (&ag)[FB]) {
#pragma unroll
    for (int kb = 0; kb < FB; ++kb) ag[kb] = csum[kb];
    unsigned bits = x.sC[jc * N + k];
    while (bits) {
      const int p = __builtin_ctz(bits);
      bits &= bits - 1;
      const float* bp = myrow + p * (FZ_TG * ROWF);
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) ag[kb] -= ld4(bp + kb * 4);
    }
  };

  // ---- stages 1..L: a_{s-1} = Agg(h_{s-1}) from LDS, h_s = act([h_{s-1} | xe | a_{s-1}] . W_s + b_s)
  for (int s = 1; s <= L; ++s) {
    float* hp = hptr(s);
    float* ap = aptr(s - 1);
    unsigned short* gp = gptr(s);
    const bool relu = s < L;
    const bool fro = a.frag_out && s == L;                       // the last stage's rows go to k_mlp_train_wg only
    f32x4 ag[NSA][FB];                     // gathered a_{s-1} rows; slot i's registers become its output h_s afterwards
    // (1) gather phase
    if (compl_sums) colsum();
    else gather_all(ag);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (compl_sums) gather_c(wv + FZ_WAVES * i, ag[i]);
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) stg4(ap + rowi[i] * F + kb * 16 + 4 * kg, ag[i][kb]);
    }
    ts.mark();                                                   // stage: gathers done
    // (2) MFMA phase.  (Not in turns as in the backward: a forward slot alone takes 3.0-3.3 us -- 1.5x the weights per
    //     slot, and the h_s stores --, side by side the two waves of a SIMD finish sooner: 53.4 us against 58.5.)
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int k = wv + FZ_WAVES * i;
      const float* cur = item_base(s, i);
      const float* nxt = i + 1 < NS ? item_base(s, i + 1) : item_base(s + 1, 0);
      f32x4 hb[FB], bias[FB], acc[FB];
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) hb[kb] = ld4(myrow + k * FZ_TG * ROWF + kb * 4);
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!RING) wload(0, cur);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (RING) wload((c + 2) % 3, c == 0 ? cur : nxt);        // refill the buffer consumed one chunk ago: 2 chunks ahead
        if (c == (NCH > 1 ? 1 : 0)) {
#pragma unroll
          for (int nt = 0; nt < FB; ++nt) bias[nt] = ldg4(cur + KB * FB * 256 + nt * 16 + 4 * kg);
        }
#pragma unroll
        for (int q = 0; q < CKB; ++q) {
          const int kb = c * CKB + q;
          const f32x4 bv = kb < FB ? hb[kb < FB ? kb : 0] : (kb == FB ? xev[i] : ag[i][kb > FB ? kb - FB - 1 : 0]);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int nt = 0; nt < FB; ++nt) acc[nt] = V2X_MFMA(wr[c][q * FB + nt][s4], bv[s4], acc[nt]);
        }
        if (RING) {
#pragma unroll
          for (int u = 0; u < CHN; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      unsigned gb = 0u;
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) {
        f32x4 v = acc[nt] + bias[nt];
        if (relu) v = relu4(v);
        ag[i][nt] = v;
        psum[nt] = i == 0 ? v : psum[nt] + v;
        gb |= gate_bits4(v) << (4 * nt);
      }
      // (two store sequences under a uniform branch, each with immediate offsets: a stride selected at run time gives
      //  every store its own 64-bit address -- measured +2 us on this kernel)
      if (fro) {
        float* hf = hp + ((int64_t)k * gridDim.x + blockIdx.x) * (FB * 256) + lane * 4;
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) stg4(hf + nt * 256, ag[i][nt]);
      } else {
        float* hr = hp + rowi[i] * F + 4 * kg;
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) stg4(hr + nt * 16, ag[i][nt]);
      }
      gp[k * gate_stride + gate_lane] = (unsigned short)gb;
      ts.mark();                                                 // slot: MFMAs + stores issued
    }
    if (compl_sums) {                      // (the partials are only read between the two barriers below)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(mysum + nt * 4, NS > 0 ? psum[nt] : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    __syncthreads();                       // every wave is done reading the h_{s-1} tile (and the totals)
    ts.mark();                                                   // stage: barrier passed
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(myrow + (wv + FZ_WAVES * i) * FZ_TG * ROWF + nt * 4, ag[i][nt]);
    }
    if (compl_sums) fz_reduce_sums<FB, ROWF>(x.sS, x.sT);
    __syncthreads();
    ts.mark();                                                   // stage: tile replaced
  }

  // ---- a_L = Agg(h_L) for the decision MLP
  {
    float* ap = aptr(L);
    auto put = [&](int i, const f32x4 (&ag)[FB]) {
      if (a.frag_out) {
        float* af = ap + ((int64_t)(wv + FZ_WAVES * i) * gridDim.x + blockIdx.x) * (FB * 256) + lane * 4;
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) stg4(af + kb * 256, ag[kb]);
      } else {
        float* ar = ap + rowi[i] * F + 4 * kg;
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) stg4(ar + kb * 16, ag[kb]);
      }
    };
    if constexpr (COMPL) {               // slot by slot: three slots' sums side by side spill next to nothing else here, but
      colsum();                          // the complement form never needs them together
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        f32x4 ag[FB];
        gather_c(wv + FZ_WAVES * i, ag);
        put(i, ag);
      }
    } else {
      f32x4 agl[NSA][FB];
      gather_all(agl);
#pragma unroll
      for (int i = 0; i < NS; ++i) put(i, agl[i]);
    }
  }
  ts.mark(true);                                                 // end
}

// SPW = ceil(N / 8) slots for the first N - 8 (SPW - 1) waves, SPW - 1 for the others
template <int F, int SPW, bool TS = false, bool COMPL = false>
__global__ __launch_bounds__(FZ_THREADS, 2) void k_gnn_fwd_fused(FusedFwdArgs a) {
  using P = FzPack<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  FzCtx x;
  x.N = a.N; x.L = a.L;
  x.SUB = a.N * FZ_TG * P::ROWF;                                 // floats per k-group sub-array
  x.sH = smem;                                                   // [4][N*16][ROWF]
  x.sS = x.sH + 4 * x.SUB;                                       // compl_sums: [8 waves][4][16][ROWF] partial column sums
  x.sT = x.sS + (COMPL ? FZ_SUMS_ROWS * P::ROWF : 0);            // compl_sums: [4][16][ROWF] their totals
  x.sRp = reinterpret_cast<int*>(x.sT + (COMPL ? FZ_TOT_ROWS * P::ROWF : 0));    // [16 N + 1] edge offsets relative to the tile
  x.sC = reinterpret_cast<unsigned*>(x.sRp + FZ_TG * a.N + 1);   // [16 N] the rows' in-neighbour sets (complement form: non-neighbours)
  x.sCol = reinterpret_cast<unsigned char*>(x.sC + FZ_TG * a.N);   // [edges] graph-local sources
  x.lane = threadIdx.x & 63;
  x.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  x.kg = x.lane >> 4;
  x.g0 = blockIdx.x * FZ_TG;
  const int ng = min(FZ_TG, a.n_graphs - x.g0);
  x.jc = min(x.lane & 15, ng - 1);
  const int r_begin = x.g0 * a.N, nrows = ng * a.N;
  if (x.wv < a.N - FZ_WAVES * (SPW - 1)) fused_fwd_body<F, SPW, TS, COMPL>(a, x, nrows, r_begin);
  else fused_fwd_body<F, SPW - 1, TS, COMPL>(a, x, nrows, r_begin);
}

// ---------------------------------------------------------------------------------------------------------------
// backward of the graph layers:  for s = L..0   dpre_s = (dh_s + Agg^T(dagg_s)) * act'(h_s),
//                                for s >= 1     [dh_{s-1} | dagg_{s-1}] = dpre_s . [W1h_s | W3_s]^T
// In: gha = [dh_L | dagg_L] (decision-MLP backward).  Out: dpre_s for the weight-gradient launch.  gha is reused as
// only read: dh_{s-1} of a row is produced and consumed by the same wave (it owns the row's slot in every stage) and
// stays in its registers, the dagg half goes from registers into the LDS tile after the barrier.  Same phase structure as the forward kernel: transposed gathers of all own slots, then the MFMAs with the
// weight fragments streaming through a register ring of 3 one-k-block chunks (2 in flight), restarted per stage.
// ---------------------------------------------------------------------------------------------------------------
struct FusedBwdArgs {
  const int32_t* row_ptr; const int32_t* col_idx;
  const float* pk;                                   // fragment-major backward weights
  const unsigned short* gate[FZ_MAXL + 1];           // ReLU' gates of h_s (FusedFwdArgs::gate), s < L
  float* dpre[FZ_MAXL + 1];
  float* gha;                                        // [R][2F]
  int n_graphs, N, L, S, edges_cap, n_edges;
  int compl_sums;                                    // see FusedFwdArgs
  const unsigned* nbmask;                            // in-neighbour sets (complement form: non-neighbour sets) left by the fused forward (FusedFwdArgs::nbmask)
  int frag_gha;                                      // gha fragment-major (written by k_mlp_train_wg, MlpArgs::frag_groups)
  int* err;
  long long* ts;
};

struct FzCtxB {
  float* sD; int* sRp; unsigned* sM; unsigned char* sCol; float* sS; float* sT; int* sFlag;
  int N, L, SUB, lane, wv, jc, kg, g0;
};

template <int F, int NS, bool TS, bool COMPL>
__device__ __forceinline__ void fused_bwd_body(const FusedBwdArgs& a, const FzCtxB& x, const int nrows, const int r_begin) {
  using P = FzPack<F>;
  constexpr int FB = P::FB, ROWF = P::ROWF;
  constexpr bool RING = FB >= 3 && NS > 0;                       // chunk = one k-block x 2FB n-tiles (8 float4, 32 MFMAs at F = 64)
  constexpr int NCH = RING ? FB : 1, CKB = FB / NCH, CHN = CKB * 2 * FB, NB = RING ? 3 : 1;
  constexpr int NSA = NS > 0 ? NS : 1;
  const int N = x.N, L = x.L, lane = x.lane, wv = x.wv, kg = x.kg, jc = x.jc;
  FzStamp<TS> ts(a.ts ? a.ts + 512 : nullptr, wv, lane);
  ts.mark();

  int64_t rowi[NSA];
#pragma unroll
  for (int i = 0; i < NS; ++i) rowi[i] = (int64_t)(x.g0 + jc) * N + (wv + FZ_WAVES * i);
  auto item_base = [&](int s, int i) -> const float* {           // stage s >= 1, this wave's i-th slot
    return a.pk + ((int64_t)(s - 1) * a.S + (a.S == 1 ? 0 : wv + FZ_WAVES * i)) * P::BWD;
  };
  f32x4 wr[NB][CHN];
  auto wload = [&](int buf, int c, const float* base) {
    gvec_p wp = (gvec_p)base + lane + c * CHN * 64;
#pragma unroll
    for (int u = 0; u < CHN; ++u) wr[buf][u] = wp[u * 64];
  };
  float* myrow = x.sD + kg * x.SUB + jc * ROWF;
  // ---- requests in the order of their urgency: CSR slice, the dagg_L / dh_L rows of the own slots
  // (complement form: no CSR at all -- the forward left the rows' non-neighbour masks behind, one coalesced word per row
  //  instead of the chain row_ptr -> col_idx slice -> LDS -> 18 LDS atomics per row that the start of this kernel waited for)
  //  Round 4: the edge form takes the same road -- the forward leaves the rows' in-neighbour sets there; the CSR walk with
  //  its atomics stays for a backward that runs without such a forward (a.nbmask null).)
  if (threadIdx.x < 4) x.sFlag[threadIdx.x] = 0;                 // (before the first barrier; stages count down from L >= 1)
  FzCsrEarly csr;
  unsigned nbv = 0u;
  const bool from_masks = COMPL || a.nbmask != nullptr;          // (uniform)
  if (from_masks) nbv = a.nbmask[r_begin + min((int)threadIdx.x, nrows - 1)];
  else csr_issue(csr, a.row_ptr, a.col_idx, r_begin, nrows, x.g0, a.edges_cap, a.n_edges);
  f32x4 dg[NSA][FB];                     // dagg rows of the own slots on their way into the LDS tile
  f32x4 dhk[NSA][FB];                    // dh rows of the own slots: produced and consumed by this wave, never leave it
  const bool frg = a.frag_gha != 0;
  const int gst = frg ? 256 : 16;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int64_t go = frg ? ((int64_t)(wv + FZ_WAVES * i) * gridDim.x + blockIdx.x) * (2 * FB * 256) + lane * 4 : rowi[i] * (2 * F) + 4 * kg;
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) {
      dg[i][nt] = ldg4(a.gha + go + (FB + nt) * gst);
      dhk[i][nt] = ldg4(a.gha + go + nt * gst);
    }
  }
  if (from_masks) {
    if ((int)threadIdx.x < FZ_TG * N) x.sRp[threadIdx.x] = (int)nbv;       // (the CSR offsets' place: unused in this form)
  } else {
    const int e_begin = a.row_ptr[r_begin], nedges = a.row_ptr[r_begin + nrows] - e_begin;
    if (nedges > a.edges_cap || nedges < 0) {
      if (threadIdx.x == 0 && a.err) atomicOr(a.err, 1);
      return;
    }
    for (int i = threadIdx.x; i < FZ_TG * N; i += FZ_THREADS) x.sM[i] = 0u;
    csr_commit(csr, a.row_ptr, a.col_idx, x.sRp, x.sCol, N, r_begin, nrows, e_begin, nedges);
  }
  constexpr bool compl_sums = COMPL;
  float* mysum = x.sS + ((wv * 4 + kg) * FZ_TG + jc) * ROWF;     // this wave's partial column sum of the dagg tile
  const float* tot = x.sT + (kg * FZ_TG + jc) * ROWF;            // column sums of the dagg tile
  auto park_sums = [&]() {               // sum of the own dagg rows (complement form); read between the stage's two barriers only
    if (compl_sums) {
      f32x4 ps[FB];
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) ps[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) ps[nt] = i == 0 ? dg[i][nt] : ps[nt] + dg[i][nt];
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(mysum + nt * 4, ps[nt]);
    }
  };
  auto park = [&]() {                    // own dagg rows into the tile
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(myrow + (wv + FZ_WAVES * i) * FZ_TG * ROWF + nt * 4, dg[i][nt]);
  };
  park_sums();
  park();
  fz_barrier();
  if (compl_sums) fz_reduce_sums<FB, ROWF>(x.sS, x.sT);
  // transposed adjacency: bit q of sM[j*N + p] = edge p -> q
  if (from_masks) {                      // from the forward's sets: p -> q exists iff bit p of row q's in-neighbour set is set
    const unsigned vmask = N >= 32 ? 0xffffffffu : (1u << N) - 1u;     // (complement form: ... of its NON-neighbour set is clear)
    for (int r = threadIdx.x; r < nrows; r += FZ_THREADS) {
      const int jj = r / N, p = r - jj * N;
      unsigned ns = 0u;
      for (int q = 0; q < N; ++q) ns |= (((unsigned)x.sRp[jj * N + q] >> p) & 1u) << q;
      x.sM[r] = COMPL ? (~ns & vmask) : ns;
    }
  } else {                               // integer atomics: order-independent
    for (int r = threadIdx.x; r < nrows; r += FZ_THREADS) {
      const int jj = r / N, q = r - jj * N;
      for (int e = x.sRp[r]; e < x.sRp[r + 1]; ++e) atomicOr(&x.sM[jj * N + x.sCol[e]], 1u << q);
    }
  }
  fz_barrier();
  ts.mark();                                                     // 1: tile + masks ready

  const unsigned valid = N >= 32 ? 0xffffffffu : (1u << N) - 1u;
  typedef const __attribute__((address_space(4))) uint64_t* CQ;
  CQ kq = (CQ)__builtin_amdgcn_kernarg_segment_ptr();
  auto gptr = [&](int s) { return reinterpret_cast<const unsigned short*>(kq[offsetof(FusedBwdArgs, gate) / 8 + s]); };
  const int gate_lane = blockIdx.x * 64 + lane, gate_stride = gridDim.x * 64;
  auto dptr = [&](int s) { return reinterpret_cast<float*>(kq[offsetof(FusedBwdArgs, dpre) / 8 + s]); };

  for (int s = L; s >= 0; --s) {
    float* dp = dptr(s);
    const bool gate = s < L;
    // (no load may be in flight across the loop's back edge or next to a store when it is waited for: hipcc then waits
    //  for vmcnt(0) -- the gates of a stage are requested at its start, one dword per slot, nothing queued in front)
    unsigned gb[NSA];
    {
      const unsigned short* gp = gptr(min(s, L - 1 > 0 ? L - 1 : 0));
#pragma unroll
      for (int i = 0; i < NSA; ++i) gb[i] = gp[(wv + FZ_WAVES * (i < NS ? i : 0)) * gate_stride + gate_lane];
    }
    // the first two weight chunks of this stage land while the gathers run
    if (RING && s > 0) { wload(0, 0, item_base(s, 0)); wload(1, 1, item_base(s, 0)); }
    f32x4 dpre[NSA][FB];
    // (1) transposed gathers (ascending destinations, two per iteration: k_agg_small<true> order), + dh, ReLU' gate
    if constexpr (!COMPL) {
      // edge form, all of the wave's slots at once (see the forward's gather_all): walk the destinations q = 0 .. N - 1, read
      // dagg row q once, add it to the slots p_i that send to q (bit q of the by-source set); ascending q = k_agg_small<true>
      unsigned msk[NSA];
#pragma unroll
      for (int i = 0; i < NSA; ++i) {
        msk[i] = i < NS ? x.sM[jc * N + wv + FZ_WAVES * i] : 0u;
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) dpre[i][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      unsigned orm = 0u;                   // degree-aware, as the forward's gather_all: sparse lanes walk their destinations only
#pragma unroll
      for (int i = 0; i < NS; ++i) orm |= msk[i];
      if (__any(2 * __popc(orm) < N)) {
        unsigned walk = 2 * __popc(orm) < N ? orm : valid;
        while (walk) {
          const int q = __builtin_ctz(walk);
          walk &= walk - 1;
          const float* bq = myrow + q * (FZ_TG * ROWF);
          f32x4 v[FB];
#pragma unroll
          for (int kb = 0; kb < FB; ++kb) v[kb] = ld4(bq + kb * 4);
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const float f = (float)((msk[i] >> q) & 1u);
#pragma unroll
            for (int kb = 0; kb < FB; ++kb) dpre[i][kb] += v[kb] * f;
          }
        }
      } else {
        for (int q = 0; q < N; ++q) {
          const float* bq = myrow + q * (FZ_TG * ROWF);
          f32x4 v[FB];
#pragma unroll
          for (int kb = 0; kb < FB; ++kb) v[kb] = ld4(bq + kb * 4);
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            const float f = (float)((msk[i] >> q) & 1u);
#pragma unroll
            for (int kb = 0; kb < FB; ++kb) dpre[i][kb] += v[kb] * f;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      f32x4 acc[FB];
      if constexpr (COMPL) {                                          // column sum minus the rows of the non-successors
        // (the column sum is re-read per slot: keeping it in registers across the slots spills next to the weight ring)
        unsigned bits = ~x.sM[jc * N + wv + FZ_WAVES * i] & valid;
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) acc[kb] = ld4(tot + kb * 4);
        while (bits) {
          const int q0 = __builtin_ctz(bits);
          bits &= bits - 1;
          const float* b0 = myrow + q0 * (FZ_TG * ROWF);
#pragma unroll
          for (int kb = 0; kb < FB; ++kb) acc[kb] -= ld4(b0 + kb * 4);
        }
      } else {
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) acc[kb] = dpre[i][kb];
      }
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) {
        acc[kb] += dhk[i][kb];
        if (gate) acc[kb] = gate_apply4(acc[kb], gb[i] >> (4 * kb));
        stg4(dp + rowi[i] * F + kb * 16 + 4 * kg, acc[kb]);
        dpre[i][kb] = acc[kb];
      }
    }
    ts.mark();                                                   // stage: gathers done
    if (s == 0) break;
#ifndef V2X_FZ_NO_TURNS
    // The two waves of a SIMD take TURNS at the MFMA phase: issuing their MFMA runs side by side they finish 5 slots in
    // 13 us, one after the other in 9.5 (a slot takes 1.9 us when its wave has the matrix pipe to itself, and the
    // second wave's gathers overlap with the first one's MFMAs either way).  Wave w + 4 waits for wave w's flag.
    if (wv >= 4) {
      while (__hip_atomic_load(x.sFlag + (wv - 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != s) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // (2) data gradients of the own slots: chunk sequence number n = i * NCH + c lives in ring buffer n % 3 and is
    //     requested two chunks ahead (inside this stage only)
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      f32x4 o[2 * FB];
#pragma unroll
      for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!RING) wload(0, 0, item_base(s, i));
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int n = i * NCH + c, nn = n + 2;
        const bool pf = RING && nn < NS * NCH;
        if (pf) wload(nn % 3, nn % NCH, item_base(s, nn / NCH));
#pragma unroll
        for (int q = 0; q < CKB; ++q) {
          const int kb = c * CKB + q;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = V2X_MFMA(wr[RING ? n % 3 : 0][q * 2 * FB + nt][s4], dpre[i][kb][s4], o[nt]);
        }
        if (pf) {
          if (n == 0) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);            // the run's first wait sees no new request
#pragma unroll
          for (int u = 0; u < CHN; ++u) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) {
        dhk[i][nt] = o[nt];                                      // dh_{s-1}
        dg[i][nt] = o[FB + nt];                                  // dagg_{s-1}: into the LDS tile after the barrier
      }
      ts.mark();                                                 // slot done
    }
#ifndef V2X_FZ_NO_TURNS
    if (wv < 4 && lane == 0) __hip_atomic_store(x.sFlag + wv, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    park_sums();
    fz_barrier();                       // all gathers from the dagg_s tile are done
    ts.mark();
    park();
    if (compl_sums) fz_reduce_sums<FB, ROWF>(x.sS, x.sT);
    fz_barrier();
    ts.mark();                                                   // stage: tile replaced
  }
  ts.mark(true);
}

template <int F, int SPW, bool TS = false, bool COMPL = false>
__global__ __launch_bounds__(FZ_THREADS, 2) void k_gnn_bwd_fused(FusedBwdArgs a) {
  using P = FzPack<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  FzCtxB x;
  x.N = a.N; x.L = a.L;
  x.SUB = a.N * FZ_TG * P::ROWF;
  x.sD = smem;                                                   // dagg tile, same layout as the forward tile
  x.sS = x.sD + 4 * x.SUB;                                       // compl_sums: partial column sums
  x.sT = x.sS + (COMPL ? FZ_SUMS_ROWS * P::ROWF : 0);            // compl_sums: their totals
  x.sRp = reinterpret_cast<int*>(x.sT + (COMPL ? FZ_TOT_ROWS * P::ROWF : 0));    // [16 N + 1]
  x.sM = reinterpret_cast<unsigned*>(x.sRp + FZ_TG * a.N + 1);   // [16 N] out-neighbour bit masks (N <= 32)
  x.sCol = reinterpret_cast<unsigned char*>(x.sM + FZ_TG * a.N);
  x.sFlag = reinterpret_cast<int*>(x.sCol + (a.edges_cap + 15) / 16 * 16);   // [4] turn flags of the SIMDs' wave pairs
  x.lane = threadIdx.x & 63;
  x.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  x.kg = x.lane >> 4;
  x.g0 = blockIdx.x * FZ_TG;
  const int ng = min(FZ_TG, a.n_graphs - x.g0);
  x.jc = min(x.lane & 15, ng - 1);
  const int r_begin = x.g0 * a.N, nrows = ng * a.N;
  if (x.wv < a.N - FZ_WAVES * (SPW - 1)) fused_bwd_body<F, SPW, TS, COMPL>(a, x, nrows, r_begin);
  else fused_bwd_body<F, SPW - 1, TS, COMPL>(a, x, nrows, r_begin);
}

}  // namespace v2x
