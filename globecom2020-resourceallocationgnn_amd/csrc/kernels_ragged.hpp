// Graph-major FUSED forward of the graph layers for VARIABLE-SIZE graphs with shared weights (BASELINE configs[4]: 8-128
// links per graph packed with CSR offsets; GNNLayer.call / AggLayer.call, BS_brain.py:44-51, :69-76, generalised as in
// SURVEY.md Appendix E).
//
// The layer-wise path runs embed, L node updates and L + 1 aggregations as 2 L + 2 launches that hand h_s / a_s to each
// other through HBM.  With shared weights every node row meets the SAME weight matrix, so an MFMA tile is ANY 16 rows of a
// workgroup's graphs -- no 16-graphs-per-slot constraint as in kernels_fused.hpp: a workgroup takes a run of whole graphs
// (<= RG_CAP rows), keeps their feature rows in LDS for all stages, aggregates out of that tile and runs the node updates
// against a weight image in LDS.  h_s and a_s are still written once each (the weight gradients, the decision MLP and the
// layer-wise backward read them); nothing is read back.
//
// Work split: the batch's rows are cut at multiples of CAPP = RG_CAP - max_nodes + 1; workgroup w owns the graphs whose FIRST
// row lies in [w CAPP, (w + 1) CAPP) -- at most RG_CAP rows, every graph exactly once, balanced by rows whatever the size
// mix.  The first graph of every interval comes from a plan k_adj_masks writes (one thread per graph: no search) -- or, by
// default, from k_ragged_plan below, which packs the same runs tightly.
// What does NOT help (round 4, built and measured at configs[4]): the tile's graphs as two halves half a stage apart -- waves
// 0-3 run the node update (MFMAs) of their graphs while waves 4-7, one per SIMD next to them, write / sum / aggregate theirs,
// plain workgroup barriers at three matched points per slot.  Taken alone the MFMA role made the launch 70 us and the other
// role 62 us; together 109 us, against 101 us for this kernel: a SIMD does not overlap one wave's fp32 MFMAs with another
// wave's vector-ALU work (it does overlap them with LDS and memory waits -- which two waves in the SAME phase already cover).
// Aggregation, per row: through the complement (column sum of the graph minus the rows at the ZERO bits of the row's
// by-destination mask) when the row has more edges than non-edges and its graph at least 16 nodes, else the rows at the ONE
// bits directly -- k_agg_dense's rule (kernels_wide.hpp); masks are the ones k_adj_masks builds for the backward anyway.
#pragma once
#include "kernels.hpp"
#include "kernels_fused.hpp"
#include "kernels_wide.hpp"
#include <cstddef>

namespace v2x {

constexpr int RG_CAP = 320;                    // rows of a workgroup's LDS tile: with graphs of <= 128 nodes a workgroup holds 193 rows
                                               // on average = 12 row tiles = 3 per SIMD (waves w and w + 4 share one: 2 + 1)
#ifndef V2X_RG_WAVES
#define V2X_RG_WAVES 8
#endif
constexpr int RG_WAVES = V2X_RG_WAVES, RG_THREADS = 64 * RG_WAVES;
constexpr int RG_RT = (RG_CAP / 16 + RG_WAVES - 1) / RG_WAVES;   // row tiles a wave may own (tiles wv, wv + 8, wv + 16)
constexpr int RG_BIG = 16;                     // graphs of >= RG_BIG nodes get a column-sum slot (at most RG_CAP / RG_BIG + 1 per tile)
constexpr int RG_SLOTS = RG_CAP / RG_BIG + 1;
constexpr int RG_MW = 4;                       // mask words per row (graphs of <= 128 nodes)

// The same split, packed: workgroup w + 1 starts at the LAST graph that still fits behind workgroup w's first one
// (rows <= RG_CAP) -- the fewest workgroups a split into runs of whole graphs can have, tiles ~ (RG_CAP - mean size / 2)
// rows full instead of CAPP on average (configs[4]: 490 workgroups of ~285 rows instead of 720 of 193 -- two rounds of the
// 256 CUs instead of three).  The chain "next start" is data dependent; one workgroup resolves it without walking it:
// nxt[g] by a bounded binary search over the offsets for every g at once, then log2(n_wgs) rounds of pointer doubling,
// workgroup w composing the powers named by the bits of w.  All in LDS (3 (n_graphs + 1) + n_wgs + 1 words: the host
// falls back to the interval plan of k_adj_masks past RG_PLAN_LDS_WORDS).  plan[w] = first graph of workgroup w, plan[w]
// = n_graphs for the launch's surplus workgroups (the grid stays the interval count, an upper bound known without a
// read-back), which leave at once.
constexpr int RG_PLAN_THREADS = 1024, RG_PLAN_LDS_WORDS = 38 * 1024;
struct RaggedPlanArgs { const int32_t* graph_off; int32_t* plan; int n_graphs, n_wgs, cap; };

__global__ __launch_bounds__(RG_PLAN_THREADS) void k_ragged_plan(RaggedPlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) int splan[];
  const int B = a.n_graphs, W = a.n_wgs, tid = threadIdx.x;
  int* off = splan;
  int* cur = off + B + 1;
  int* oth = cur + B + 1;
  int* first = oth + B + 1;                                                // [W + 1]
  for (int g = tid; g <= B; g += RG_PLAN_THREADS) off[g] = a.graph_off[g];
  __syncthreads();
  rg_plan_tables<int, RG_PLAN_THREADS>(off, cur, oth, first, B, W, a.cap, a.plan, tid);
}

// One lane's share of a graph's column sum: rows rg, rg + RGN, ... of the tile, four loads in flight and four running sums
// (a 128-node graph was a chain of 32 dependent LDS reads); fixed order, so the sums stay reproducible.
template <int LDT, int RGN>
__device__ __forceinline__ f32x4 rg_column_partial(const float* col, int rg, int n) {
  f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int r = rg;
  for (; r + 3 * RGN < n; r += 4 * RGN) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(col + r * LDT), v1 = *reinterpret_cast<const f32x4*>(col + (r + RGN) * LDT),
                v2 = *reinterpret_cast<const f32x4*>(col + (r + 2 * RGN) * LDT), v3 = *reinterpret_cast<const f32x4*>(col + (r + 3 * RGN) * LDT);
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; r < n; r += RGN) s0 += *reinterpret_cast<const f32x4*>(col + r * LDT);
  return (s0 + s1) + (s2 + s3);
}

// Per-row record of the aggregation walk (built once per workgroup, read by every stage): the first K entries of the
// row's walk -- the ZERO bits of its mask if the row goes through the complement, else the ONE bits (k_agg_dense's rule:
// complement when the graph has a column-sum slot and the row more edges than non-edges) -- 7 bits each, their count, and
// whether the walk has more.  With the reference topology a destination's complement is itself and the link it does not
// hear (2 entries: the forward keeps 3 in one word).  The backward (a SOURCE's complement is itself and the links that do not
// hear it, 1 + Poisson(1) entries) keeps the bit walk: its 254 registers have no room for several rows in flight -- a
// two-word record with the gates held as bits spilled and cost 104 -> 138 us.
constexpr int RG_REC_K = 3, RG_REC_CNT = 21;
constexpr unsigned RG_REC_MORE = 1u << 23, RG_REC_DIRECT = 1u << 24;
template <int K>
__device__ __forceinline__ void rg_row_entries(const unsigned* mrow, int n, int slot, unsigned (&ent)[K], int& cnt, bool& more,
                                               bool& direct) {
  unsigned z[RG_MW];
  int ones = 0;
#pragma unroll
  for (int w = 0; w < RG_MW; ++w) {
    const int left = n - 32 * w;
    const unsigned lim = left <= 0 ? 0u : (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
    z[w] = mrow[w] & lim;
    ones += __builtin_popcount(z[w]);
  }
  direct = slot >= RG_SLOTS || 2 * ones < n;
  more = false;
  cnt = 0;
#pragma unroll
  for (int e = 0; e < K; ++e) ent[e] = 0u;
#pragma unroll
  for (int w = 0; w < RG_MW; ++w) {
    const int left = n - 32 * w;
    const unsigned lim = left <= 0 ? 0u : (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
    unsigned zz = direct ? z[w] : ~z[w] & lim;
    while (zz && cnt < K) {
      const unsigned p = 32 * w + __builtin_ctz(zz);
#pragma unroll
      for (int e = 0; e < K; ++e)
        if (e == cnt) ent[e] = p;
      zz &= zz - 1;
      ++cnt;
    }
    if (zz) more = true;
  }
}
__device__ __forceinline__ unsigned rg_row_record(const unsigned* mrow, int n, int slot) {
  unsigned ent[RG_REC_K];
  int cnt; bool more, direct;
  rg_row_entries<RG_REC_K>(mrow, n, slot, ent, cnt, more, direct);
  unsigned rec = (direct ? RG_REC_DIRECT : 0u) | (more ? RG_REC_MORE : 0u) | (unsigned)cnt << RG_REC_CNT;
#pragma unroll
  for (int e = 0; e < RG_REC_K; ++e) rec |= ent[e] << (7 * e);
  return rec;
}
struct RaggedFwdArgs {
  const float* xe; const int32_t* graph_off; const int32_t* row_ptr;
  const unsigned* adjT;                        // [R][mask_words] by destination: bit p of adjT[q] = edge p -> q
  const int32_t* plan;                         // [n_wgs + 1]: first graph of every row interval (k_adj_masks)
  const float* W[FZ_MAXL + 1];                 // stage s weights [K][F] row-major + bias (flat parameter layout, shared slot)
  float* h[FZ_MAXL + 1]; float* a[FZ_MAXL + 1];
  int n_graphs, n_rows, L, mask_words, capp, xr;   // xr = real rows of the packed [x | e] block
  int* err;
  long long* ts;                               // measurement: [8 waves][64] 100 MHz time stamps of workgroup 100 (V2X_FUSED_TS=1)
};

template <int F>
struct RaggedLds {
  // tile rows of F + 4 floats: a quarter-wave of a b128 access (16 rows j, one k-group) covers all 64 banks
  static constexpr int FB = F / 16, LDT = F + 4;
  static constexpr int KP = 2 * F + XE, LDW = F + 4;
  static constexpr int TILE = 0, SUMS = TILE + RG_CAP * LDT, WIMG = SUMS + RG_SLOTS * LDT, BIAS = WIMG + KP * LDW, MASK = BIAS + F,
                       INFO = MASK + RG_CAP * RG_MW, GOFF = INFO + RG_CAP, REC = GOFF + RG_CAP + 8, TOTAL = REC + RG_CAP;
};

template <int F>
__global__ __launch_bounds__(RG_THREADS, 1) void k_gnn_fwd_ragged(RaggedFwdArgs a) {
  using Lds = RaggedLds<F>;
  constexpr int FB = Lds::FB, LDT = Lds::LDT, KP = Lds::KP, LDW = Lds::LDW;
  constexpr int WP = (KP * (F / 4) + RG_THREADS - 1) / RG_THREADS;        // float4 passes of a weight image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem + Lds::TILE; float* sS = smem + Lds::SUMS; float* sW = smem + Lds::WIMG; float* sBias = smem + Lds::BIAS;
  unsigned* sMask = reinterpret_cast<unsigned*>(smem + Lds::MASK);
  int* sInfo = reinterpret_cast<int*>(smem + Lds::INFO);                  // per row: r0 | n << 9 | slot << 17  (slot = RG_SLOTS: none)
  int* sGoff = reinterpret_cast<int*>(smem + Lds::GOFF);                  // the tile's graph offsets (local rows)
  unsigned* sRec = reinterpret_cast<unsigned*>(smem + Lds::REC);          // per row: rg_row_record
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  // per-stage pointers through the kernarg segment (a by-value array indexed at run time would be copied to scratch)
  typedef const __attribute__((address_space(4))) unsigned char* CBytes;
  const CBytes kargs = (CBytes)__builtin_amdgcn_kernarg_segment_ptr();
  auto stage_ptr = [&](size_t field_off, int s) -> float* {
    return *reinterpret_cast<float* const __attribute__((address_space(4)))*>(kargs + field_off + 8 * (size_t)s);
  };
  int n_ts = 0;
  auto stamp = [&]() {
    if (a.ts && blockIdx.x == 100 && lane == 0 && n_ts < 64) a.ts[wv * 64 + n_ts] = (long long)wall_clock64();
    ++n_ts;
  };
  stamp();
  const int g0 = a.plan[blockIdx.x], g1 = a.plan[blockIdx.x + 1];
  if (g1 <= g0) return;
  const int R0 = a.graph_off[g0], rows = a.graph_off[g1] - R0, ng = g1 - g0;
  if (rows > RG_CAP || ng > RG_CAP) { if (tid == 0 && a.err) atomicOr(a.err, 1); return; }

  // ---- prologue: graph bounds, per-row records, masks, this lane's [x | e] fragments, the embed layer's weights
  for (int i = tid; i <= ng; i += RG_THREADS) sGoff[i] = a.graph_off[g0 + i] - R0;
  for (int i = tid; i < rows * a.mask_words; i += RG_THREADS) {
    const int r = i / a.mask_words, w = i - r * a.mask_words;
    sMask[r * RG_MW + w] = a.adjT[(int64_t)(R0 + r) * a.mask_words + w];
  }
  int rr[RG_RT];
  f32x4 xef[RG_RT];
#pragma unroll
  for (int t = 0; t < RG_RT; ++t) {
    rr[t] = 16 * (wv + RG_WAVES * t) + j;
    xef[t] = ld4(a.xe + (int64_t)(R0 + min(rr[t], rows - 1)) * XE + 4 * kg);
  }
  auto load_weights = [&](int s, float4 (&v)[WP]) {      // all loads unconditional from valid addresses, masked afterwards
    const float* Wg = stage_ptr(offsetof(RaggedFwdArgs, W), s);
    const RowPad pad = s == 0 ? RowPad{a.xr, XE - a.xr, a.xr + F} : RowPad{F + a.xr, XE - a.xr, 2 * F + a.xr};
    const int kp = s == 0 ? XE : KP;
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int i = tid + RG_THREADS * p, rp = i / (F / 4), c = (i - rp * (F / 4)) << 2;
      const int rl = rp < kp ? real_row(pad, rp) : -1;
      const bool ok = rl >= 0 && (s > 0 || rl < a.xr);          // embed: the neighbour-init rows are absent (zero input)
      const float4 t = *reinterpret_cast<const float4*>(Wg + (ok ? (int64_t)rl * F + c : 0));
      const float mk = ok ? 1.f : 0.f;
      v[p] = make_float4(t.x * mk, t.y * mk, t.z * mk, t.w * mk);
    }
  };
  auto store_weights = [&](int s, const float4 (&v)[WP]) {
    const int kp = s == 0 ? XE : KP;
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int i = tid + RG_THREADS * p, rp = i / (F / 4), c = (i - rp * (F / 4)) << 2;
      if (rp < kp) *reinterpret_cast<float4*>(sW + rp * LDW + c) = v[p];
    }
    const int k_real = s == 0 ? a.xr + F : 2 * F + a.xr;
    if (tid < F) sBias[tid] = stage_ptr(offsetof(RaggedFwdArgs, W), s)[(int64_t)k_real * F + tid];
  };
  float4 wreg[WP];
  load_weights(0, wreg);
  stamp();
  __syncthreads();                                                        // sGoff visible
  stamp();
  if (tid < rows) {                                                       // a thread per row: its graph by bisection of the tile's offsets
    int lo = 0, hi = ng;                                                  // invariant: sGoff[lo] <= tid < sGoff[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (sGoff[mid] <= tid) lo = mid; else hi = mid;
    }
    const int r0 = sGoff[lo], n = sGoff[lo + 1] - r0;
    int slot = RG_SLOTS;
    if (n >= RG_BIG) {                                                    // rank among the tile's big graphs (<= RG_SLOTS - 1 of them)
      slot = 0;
      for (int g = 0; g < lo; ++g) slot += (sGoff[g + 1] - sGoff[g]) >= RG_BIG ? 1 : 0;
    }
    if ((n < 1 || n > 128) && a.err) atomicOr(a.err, 1);
    sInfo[tid] = r0 | (n << 9) | (slot << 17);
    sRec[tid] = rg_row_record(sMask + tid * RG_MW, min(max(n, 0), 128), slot);
  }
  store_weights(0, wreg);
  __syncthreads();
  stamp();

  const int n_my = 16 * (wv + 2 * RG_WAVES) < rows ? 3 : (16 * (wv + RG_WAVES) < rows ? 2 : 1);   // this wave's row tiles
  f32x4 agg[RG_RT][FB];                                                   // a_{s-1} of this lane's rows (B operand of stage s)
#pragma unroll
  for (int t = 0; t < RG_RT; ++t)
#pragma unroll
    for (int b = 0; b < FB; ++b) agg[t][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int s = 0; s <= a.L; ++s) {
    // ---- node update of stage s for row tiles wv and wv + 8:  out = act([h_{s-1} | x e | a_{s-1}] W_s + b_s)
    f32x4 acc[RG_RT][FB];
#pragma unroll
    for (int t = 0; t < RG_RT; ++t)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (a wave skips the tiles that lie past the workgroup's rows: uniform branches; the weight values of a k-step are read
    //  once for all of the wave's tiles)
    auto kblock = [&](int kb, const f32x4 (&b)[RG_RT]) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float w[FB];
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) w[nt] = sW[(kb * 16 + 4 * kg + st) * LDW + nt * 16 + j];
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) acc[0][nt] = V2X_MFMA(w[nt], b[0][st], acc[0][nt]);
#pragma unroll
        for (int t = 1; t < RG_RT; ++t)
          if (n_my > t) {
#pragma unroll
            for (int nt = 0; nt < FB; ++nt) acc[t][nt] = V2X_MFMA(w[nt], b[t][st], acc[t][nt]);
          }
      }
    };
    if (s == 0) {
      kblock(0, xef);
    } else {
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) {
        f32x4 b[RG_RT];
#pragma unroll
        for (int t = 0; t < RG_RT; ++t) b[t] = *reinterpret_cast<const f32x4*>(sT + min(rr[t], RG_CAP - 1) * LDT + 16 * kb + 4 * kg);
        kblock(kb, b);
      }
      kblock(FB, xef);
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) {
        f32x4 b[RG_RT];
#pragma unroll
        for (int t = 0; t < RG_RT; ++t) b[t] = agg[t][kb];
        kblock(FB + 1 + kb, b);
      }
    }
    // lane holds out[row rr[t]][nt * 16 + 4 kg .. + 3]
    float* h_out = stage_ptr(offsetof(RaggedFwdArgs, h), s);
    float* a_out = stage_ptr(offsetof(RaggedFwdArgs, a), s);
#pragma unroll
    for (int t = 0; t < RG_RT; ++t)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) {
        f32x4 v = acc[t][nt] + ld4(sBias + nt * 16 + 4 * kg);
        if (s < a.L) v = relu4(v);
        acc[t][nt] = v;
        if (rr[t] < rows) st4(h_out + (int64_t)(R0 + rr[t]) * F + nt * 16 + 4 * kg, v);
      }
    stamp();
    if (s < a.L) load_weights(s + 1, wreg);                               // travels while the tile is replaced and aggregated
    __syncthreads();                                                      // everybody is done with h_{s-1} and W_s
    stamp();
#pragma unroll
    for (int t = 0; t < RG_RT; ++t)
      if (rr[t] < RG_CAP) {
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) *reinterpret_cast<f32x4*>(sT + rr[t] * LDT + 16 * nt + 4 * kg) = acc[t][nt];
      }
    if (s < a.L) store_weights(s + 1, wreg);
    stamp();
    __syncthreads();                                                      // tile = h_s
    stamp();
    // ---- column sums of the big graphs: slot k by wave k mod 8; lane = (feature float4 (kg', kb'), row group)
    {
      constexpr int COMB = 4 * FB, RGN = 64 / COMB;
      const int cb = lane % COMB, rg = lane / COMB, ckg = cb / FB, ckb = cb - ckg * FB;
      int slot = 0;
      for (int g = 0; g < ng; ++g) {
        const int r0 = sGoff[g], n = sGoff[g + 1] - r0;
        if (n < RG_BIG) continue;
        if ((slot & (RG_WAVES - 1)) == wv) {
          const f32x4 sum = rg_column_partial<LDT, RGN>(sT + r0 * LDT + 16 * ckb + 4 * ckg, rg, n);
          f32x4 tot = sum;
#pragma unroll
          for (int o = COMB; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[e] += __shfl_xor(tot[e], o, 64);
          if (rg == 0) *reinterpret_cast<f32x4*>(sS + slot * LDT + 16 * ckb + 4 * ckg) = tot;
        }
        ++slot;
      }
    }
    stamp();
    __syncthreads();
    stamp();
    // ---- a_s of this lane's rows: out of the tile, into registers (next stage's B operand) and to HBM.  A row whose walk
    // has at most RG_REC_K entries takes them from its record (prologue): all tile reads of the row are in flight at once
    // instead of one dependent LDS round trip per bit (find the bit, read, add: ~0.7 us of a ~1.5 us tile), absent entries
    // read row 0 of the graph with weight 0.  Same additions in the same order as the walk; longer rows take the walk.
#pragma unroll
    for (int t = 0; t < RG_RT; ++t) {
      f32x4 acc2[FB];
      const bool live = rr[t] < rows;
      const int row = min(rr[t], RG_CAP - 1);
      const int info = sInfo[row];
      const unsigned rec = live ? sRec[row] : 0u;
      const int r0 = live ? info & 511 : 0, slot = info >> 17;
      {
        const float* base = sT + r0 * LDT + 4 * kg;
        f32x4 v[RG_REC_K][FB], sm[FB];
#pragma unroll
        for (int e = 0; e < RG_REC_K; ++e)
#pragma unroll
          for (int b = 0; b < FB; ++b) v[e][b] = *reinterpret_cast<const f32x4*>(base + ((rec >> (7 * e)) & 127u) * LDT + 16 * b);
        const int sl = (rec & RG_REC_DIRECT) || !live ? 0 : slot;
#pragma unroll
        for (int b = 0; b < FB; ++b) sm[b] = *reinterpret_cast<const f32x4*>(sS + sl * LDT + 16 * b + 4 * kg);
        const int cnt = (rec >> RG_REC_CNT) & 3u;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
          f32x4 x = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < RG_REC_K; ++e) x += v[e][b] * (cnt > e ? 1.f : 0.f);
          acc2[b] = (rec & RG_REC_DIRECT) || !live ? x : sm[b] - x;
        }
      }
      if (rec & RG_REC_MORE) {                                            // (rows of a live lane only)
        const int n = (info >> 9) & 255;
        const bool direct = rec & RG_REC_DIRECT;
#pragma unroll
        for (int b = 0; b < FB; ++b) acc2[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < RG_MW; ++w) {
          const int left = n - 32 * w;
          if (left <= 0) break;
          const unsigned m = sMask[row * RG_MW + w];
          unsigned z = (direct ? m : ~m) & (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
          while (z) {
            const int p = 32 * w + __builtin_ctz(z);
            z &= z - 1;
            const float* src = sT + (r0 + p) * LDT + 4 * kg;
#pragma unroll
            for (int b = 0; b < FB; ++b) acc2[b] += *reinterpret_cast<const f32x4*>(src + 16 * b);
          }
        }
        if (!direct) {
#pragma unroll
          for (int b = 0; b < FB; ++b) acc2[b] = *reinterpret_cast<const f32x4*>(sS + slot * LDT + 16 * b + 4 * kg) - acc2[b];
        }
      }
      if (live) {
#pragma unroll
        for (int b = 0; b < FB; ++b) st4(a_out + (int64_t)(R0 + rr[t]) * F + b * 16 + 4 * kg, acc2[b]);
      }
#pragma unroll
      for (int b = 0; b < FB; ++b) agg[t][b] = acc2[b];
    }
    stamp();
    // (the next stage reads the tile and sW after the barrier at its own end of MFMAs; its B operands of h come from the
    //  tile, which nobody writes before that barrier)
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// The backward of the same layers in one launch: L + 1 transposed aggregations and L data gradients.  Same work split, same
// LDS tile (now the dagg_s rows of the workgroup's graphs), same per-row rule for the aggregation -- walked over the masks BY
// SOURCE (bit q of adj[p] = edge p -> q):
//     dpre_s[p] = (dh_s[p] + sum_{q : p -> q} dagg_s[q]) * relu'(h_s[p])             (no gate at s = L: h_L is linear)
//     [dh_{s-1} | dagg_{s-1}] = dpre_s . [W_s(h rows) | W_s(agg rows)]^T              (MFMA against a transposed LDS image)
// dh never leaves the lane that owns the row, dagg goes from registers into the tile; dpre_s is written once per stage for
// the weight-gradient launch.  Input: gha = [dh_L | dagg_L] of the decision MLP's backward, row-major [R][2F].
struct RaggedBwdArgs {
  const float* gha; const int32_t* graph_off;
  const unsigned* adj;                         // [R][mask_words] by source
  const int32_t* plan;
  const float* W[FZ_MAXL + 1];                 // W[s], s >= 1 (flat parameter layout, shared slot)
  const float* h[FZ_MAXL + 1];                 // h_s: the ReLU' gates (s < L)
  float* dpre[FZ_MAXL + 1];
  int n_graphs, n_rows, L, mask_words, capp, xr;
  int* err;
};

template <int F>
struct RaggedBwdLds {
  static constexpr int FB = F / 16, LDT = F + 4, LDWT = 2 * F + 4;
  static constexpr int TILE = 0, SUMS = TILE + RG_CAP * LDT, WIMG = SUMS + RG_SLOTS * LDT, MASK = WIMG + F * LDWT,
                       INFO = MASK + RG_CAP * RG_MW, GOFF = INFO + RG_CAP, TOTAL = GOFF + RG_CAP + 8;
};

template <int F>
__global__ __launch_bounds__(RG_THREADS, 1) void k_gnn_bwd_ragged(RaggedBwdArgs a) {
  using Lds = RaggedBwdLds<F>;
  constexpr int FB = Lds::FB, LDT = Lds::LDT, LDWT = Lds::LDWT;
  constexpr int WP = (2 * F * (F / 4) + RG_THREADS - 1) / RG_THREADS;     // float4 passes over the 2F weight rows of a stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem + Lds::TILE; float* sS = smem + Lds::SUMS; float* sW = smem + Lds::WIMG;
  unsigned* sMask = reinterpret_cast<unsigned*>(smem + Lds::MASK);
  int* sInfo = reinterpret_cast<int*>(smem + Lds::INFO);
  int* sGoff = reinterpret_cast<int*>(smem + Lds::GOFF);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  typedef const __attribute__((address_space(4))) unsigned char* CBytes;
  const CBytes kargs = (CBytes)__builtin_amdgcn_kernarg_segment_ptr();
  auto stage_ptr = [&](size_t field_off, int s) -> float* {
    return *reinterpret_cast<float* const __attribute__((address_space(4)))*>(kargs + field_off + 8 * (size_t)s);
  };
  const int g0 = a.plan[blockIdx.x], g1 = a.plan[blockIdx.x + 1];
  if (g1 <= g0) return;
  const int R0 = a.graph_off[g0], rows = a.graph_off[g1] - R0, ng = g1 - g0;
  if (rows > RG_CAP || ng > RG_CAP) { if (tid == 0 && a.err) atomicOr(a.err, 1); return; }

  for (int i = tid; i <= ng; i += RG_THREADS) sGoff[i] = a.graph_off[g0 + i] - R0;
  for (int i = tid; i < rows * a.mask_words; i += RG_THREADS) {
    const int r = i / a.mask_words, w = i - r * a.mask_words;
    sMask[r * RG_MW + w] = a.adj[(int64_t)(R0 + r) * a.mask_words + w];
  }
  // this lane's rows: dh_L into registers, dagg_L into the tile
  int rr[RG_RT];
  f32x4 dh[RG_RT][FB];
#pragma unroll
  for (int t = 0; t < RG_RT; ++t) {
    rr[t] = 16 * (wv + RG_WAVES * t) + j;
    const float* g = a.gha + (int64_t)(R0 + min(rr[t], rows - 1)) * (2 * F) + 4 * kg;
#pragma unroll
    for (int b = 0; b < FB; ++b) dh[t][b] = ld4(g + 16 * b);
    if (rr[t] < rows) {
#pragma unroll
      for (int b = 0; b < FB; ++b) *reinterpret_cast<f32x4*>(sT + rr[t] * LDT + 16 * b + 4 * kg) = ld4(g + F + 16 * b);
    }
  }
  // transposed image of stage s: sW[f][n] = W_s[row(n)][f], row(n) = n (h rows) for n < F, F + xr + (n - F) (agg rows) beyond
  auto load_weights = [&](int s, float4 (&v)[WP]) {
    const float* Wg = stage_ptr(offsetof(RaggedBwdArgs, W), s);
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int i = min(tid + RG_THREADS * p, 2 * F * (F / 4) - 1), n = i / (F / 4), c = (i - n * (F / 4)) << 2;
      v[p] = *reinterpret_cast<const float4*>(Wg + (int64_t)(n < F ? n : n + a.xr) * F + c);
    }
  };
  auto store_weights = [&](const float4 (&v)[WP]) {
#pragma unroll
    for (int p = 0; p < WP; ++p) {
      const int i = tid + RG_THREADS * p, n = i / (F / 4), c = (i - n * (F / 4)) << 2;
      if (i < 2 * F * (F / 4)) {
        sW[(c + 0) * LDWT + n] = v[p].x; sW[(c + 1) * LDWT + n] = v[p].y;
        sW[(c + 2) * LDWT + n] = v[p].z; sW[(c + 3) * LDWT + n] = v[p].w;
      }
    }
  };
  float4 wreg[WP];
  if (a.L >= 1) load_weights(a.L, wreg);
  __syncthreads();
  if (tid < rows) {
    int lo = 0, hi = ng;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (sGoff[mid] <= tid) lo = mid; else hi = mid;
    }
    const int r0 = sGoff[lo], n = sGoff[lo + 1] - r0;
    int slot = RG_SLOTS;
    if (n >= RG_BIG) {
      slot = 0;
      for (int g = 0; g < lo; ++g) slot += (sGoff[g + 1] - sGoff[g]) >= RG_BIG ? 1 : 0;
    }
    if ((n < 1 || n > 128) && a.err) atomicOr(a.err, 1);
    sInfo[tid] = r0 | (n << 9) | (slot << 17);
  }
  if (a.L >= 1) store_weights(wreg);
  __syncthreads();                                                        // tile = dagg_L, records, masks, W_L image
  const int n_my = 16 * (wv + 2 * RG_WAVES) < rows ? 3 : (16 * (wv + RG_WAVES) < rows ? 2 : 1);

#pragma unroll 1
  for (int s = a.L; s >= 0; --s) {
    // the first tile's gates travel while the column sums are formed (the next tile's while a tile is worked on: all of
    // them up front, next to dh, the new dagg and a tile's accumulators, did not fit the 256 registers of a wave)
    const float* hs = stage_ptr(offsetof(RaggedBwdArgs, h), s);
    auto load_gate = [&](int t, f32x4 (&g)[FB]) {
#pragma unroll
      for (int b = 0; b < FB; ++b) g[b] = ld4(hs + (int64_t)(R0 + min(rr[t], rows - 1)) * F + 16 * b + 4 * kg);
    };
    f32x4 gate_next[FB];
    if (s < a.L) load_gate(0, gate_next);
    {
      constexpr int COMB = 4 * FB, RGN = 64 / COMB;
      const int cb = lane % COMB, rg = lane / COMB, ckg = cb / FB, ckb = cb - ckg * FB;
      int slot = 0;
      for (int g = 0; g < ng; ++g) {
        const int r0 = sGoff[g], n = sGoff[g + 1] - r0;
        if (n < RG_BIG) continue;
        if ((slot & (RG_WAVES - 1)) == wv) {
          const f32x4 sum = rg_column_partial<LDT, RGN>(sT + r0 * LDT + 16 * ckb + 4 * ckg, rg, n);
          f32x4 tot = sum;
#pragma unroll
          for (int o = COMB; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[e] += __shfl_xor(tot[e], o, 64);
          if (rg == 0) *reinterpret_cast<f32x4*>(sS + slot * LDT + 16 * ckb + 4 * ckg) = tot;
        }
        ++slot;
      }
    }
    __syncthreads();
    float* dpre_out = stage_ptr(offsetof(RaggedBwdArgs, dpre), s);
    f32x4 dagg_new[RG_RT][FB];
#pragma unroll
    for (int t = 0; t < RG_RT; ++t) {
      // ---- dpre_s of the lane's row of tile t ...
      f32x4 gate[FB];
#pragma unroll
      for (int b = 0; b < FB; ++b) gate[b] = gate_next[b];
      if (s < a.L && t + 1 < RG_RT) load_gate(t + 1, gate_next);
      f32x4 dpre[FB];
#pragma unroll
      for (int b = 0; b < FB; ++b) dpre[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rr[t] < rows) {
        const int info = sInfo[rr[t]], r0 = info & 511, n = (info >> 9) & 255, slot = info >> 17;
        int ones = 0;
#pragma unroll
        for (int w = 0; w < RG_MW; ++w) {
          const int left = n - 32 * w;
          if (left > 0) ones += __builtin_popcount(sMask[rr[t] * RG_MW + w] & (left >= 32 ? 0xffffffffu : ((1u << left) - 1u)));
        }
        const bool direct = slot >= RG_SLOTS || 2 * ones < n;
#pragma unroll
        for (int w = 0; w < RG_MW; ++w) {
          const int left = n - 32 * w;
          if (left <= 0) break;
          const unsigned m = sMask[rr[t] * RG_MW + w];
          unsigned z = (direct ? m : ~m) & (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
          while (z) {
            const int q = 32 * w + __builtin_ctz(z);
            z &= z - 1;
            const float* src = sT + (r0 + q) * LDT + 4 * kg;
#pragma unroll
            for (int b = 0; b < FB; ++b) dpre[b] += *reinterpret_cast<const f32x4*>(src + 16 * b);
          }
        }
        if (!direct) {
#pragma unroll
          for (int b = 0; b < FB; ++b) dpre[b] = *reinterpret_cast<const f32x4*>(sS + slot * LDT + 16 * b + 4 * kg) - dpre[b];
        }
#pragma unroll
        for (int b = 0; b < FB; ++b) {
          f32x4 v = dpre[b] + dh[t][b];
          if (s < a.L) v = gate4(v, gate[b]);
          dpre[b] = v;
          st4(dpre_out + (int64_t)(R0 + rr[t]) * F + 16 * b + 4 * kg, v);
        }
      }
      if (s == 0) continue;
      // ---- ... and [dh_{s-1} | dagg_{s-1}] of the tile = dpre_s . Wt_s
      f32x4 acc[2 * FB];
#pragma unroll
      for (int nt = 0; nt < 2 * FB; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n_my > t) {
#pragma unroll
        for (int kb = 0; kb < FB; ++kb)
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            float w[2 * FB];
#pragma unroll
            for (int nt = 0; nt < 2 * FB; ++nt) w[nt] = sW[(kb * 16 + 4 * kg + st) * LDWT + nt * 16 + j];
#pragma unroll
            for (int nt = 0; nt < 2 * FB; ++nt) acc[nt] = V2X_MFMA(w[nt], dpre[kb][st], acc[nt]);
          }
      }
#pragma unroll
      for (int b = 0; b < FB; ++b) { dh[t][b] = acc[b]; dagg_new[t][b] = acc[FB + b]; }
    }
    if (s == 0) break;
    if (s > 1) load_weights(s - 1, wreg);
    __syncthreads();                                                      // everybody is done with dagg_s and Wt_s
#pragma unroll
    for (int t = 0; t < RG_RT; ++t)
      if (rr[t] < RG_CAP) {
#pragma unroll
        for (int b = 0; b < FB; ++b) *reinterpret_cast<f32x4*>(sT + rr[t] * LDT + 16 * b + 4 * kg) = dagg_new[t][b];
      }
    if (s > 1) store_weights(wreg);
    __syncthreads();                                                      // tile = dagg_{s-1}
  }
}

}  // namespace v2x
