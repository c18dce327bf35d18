// SMALL-TILE variants of the graph-major fused ragged kernels (kernels_ragged.hpp; BASELINE configs[4]: 8-128 links per graph, shared
// weights; GNNLayer.call / AggLayer.call, BS_brain.py:44-51, :69-76, generalised as in SURVEY.md Appendix E).
//
// k_gnn_fwd_ragged / k_gnn_bwd_ragged hold one 8-wave workgroup per CU: tile (320 rows, 85 KB) + weight image (38 / 33 KB) +
// records fill the LDS, and a workgroup alternates ~10 us of MFMAs with ~5 us of barriers, tile writes, column sums and
// aggregation with nothing else resident (0.30 / 0.32 of their roofs, WAIT_ANY 51 / 39 %).  More waves in the SAME
// workgroup do not help (12 / 16 waves: spills, profiles/r05_ragged_waves_ab.txt).  Here
//   * the weight image is NOT in LDS: with shared weights every workgroup streams the SAME fragment-major copy (k_pack_weights:
//     37 KB per stage) from L2 / L1 straight into MFMA A fragments, one k-block ahead, each fragment serving all of the wave's
//     row tiles (one 1 KiB load per 4 x tiles MFMAs);
//   * the tile is RGS_CAP = 160 rows (43.5 KB) and a workgroup 4 waves, so THREE workgroups fit a CU (51 KB of LDS, <= 168
//     registers each): one workgroup's barrier / sum / aggregation phases run under the others' MFMAs;
//   * the stage's MFMA section is instantiated per number of row tiles the wave owns (1..3) and picked once per stage: straight-line
//     code between a fragment's request and its use, so that hipcc counts its waits.
// MEASURED (round 5, configs[4] share, profiles/r05_ragged_small_ab.txt): forward 123 us against 101 us, backward 161 us (2 or 3
// workgroups per CU alike; the three-per-CU build spills) against 103 us, and the plan no longer fits the mask launch's LDS
// (+ 19 us as a launch of its own: the grid's upper bound is 4,219 runs instead of 721).  The hypothesis -- the large-tile
// kernels are starved of co-resident workgroups -- does not hold in this form: OFF by default (V2X_RAGGED_SMALL=1), kept as the
// measured alternative and as a second implementation the parity tests can run.
// Same per-row arithmetic in the same order as the large-tile kernels (a row's results do not depend on where its graph sits in
// a tile or on who owns its row tile): bitwise their values.  The device-side plan (k_ragged_plan / the mask launch's workgroup 0)
// packs runs of whole graphs of <= RGS_CAP rows instead of <= RG_CAP.
#pragma once
#include "kernels_ragged.hpp"

namespace v2x {

constexpr int RGS_CAP = 160, RGS_WAVES = 4, RGS_THREADS = 64 * RGS_WAVES;
constexpr int RGS_RT = (RGS_CAP / 16 + RGS_WAVES - 1) / RGS_WAVES;      // 3: row tiles wv, wv + 4, wv + 8
constexpr int RGS_SLOTS = RGS_CAP / RG_BIG + 1;                        // column-sum slots of a tile (the "no slot" mark stays RG_SLOTS)

template <int F>
struct RaggedSmallLds {
  static constexpr int FB = F / 16, LDT = F + 4;
  static constexpr int TILE = 0, SUMS = TILE + RGS_CAP * LDT, MASK = SUMS + RGS_SLOTS * LDT, INFO = MASK + RGS_CAP * RG_MW,
                       GOFF = INFO + RGS_CAP, REC = GOFF + RGS_CAP + 8, TOTAL = REC + RGS_CAP;
};

// node update of one stage for NMY row tiles of this wave: acc[t][nt] += sum over the stage's k-blocks, A fragments from the
// fragment-major copy (`item`: [k-block][n-tile][64 lanes][4]), one k-block ahead
template <int F, int NMY, bool EMBED>
__device__ __forceinline__ void rgs_stage_mfma(const float* item, int lane, int kg, const float* sT, const int (&rr)[RGS_RT],
                                               const f32x4 (&xef)[RGS_RT], const f32x4 (&agg)[RGS_RT][F / 16], f32x4 (&acc)[RGS_RT][F / 16]) {
  constexpr int FB = F / 16, LDT = F + 4, NKB = EMBED ? 1 : 2 * FB + 1;
  gvec_p wp = (gvec_p)item + lane;
  f32x4 w[2][FB];
#pragma unroll
  for (int nt = 0; nt < FB; ++nt) w[0][nt] = wp[nt * 64];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if (kb + 1 < NKB) {
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) w[(kb + 1) & 1][nt] = wp[((kb + 1) * FB + nt) * 64];
    }
    f32x4 b[NMY];
#pragma unroll
    for (int t = 0; t < NMY; ++t) {
      if (EMBED || kb == FB) b[t] = xef[t];
      else if (kb < FB) b[t] = *reinterpret_cast<const f32x4*>(sT + min(rr[t], RGS_CAP - 1) * LDT + 16 * kb + 4 * kg);
      else b[t] = agg[t][kb - FB - 1 < 0 ? 0 : kb - FB - 1];
    }
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt)
#pragma unroll
        for (int t = 0; t < NMY; ++t) acc[t][nt] = V2X_MFMA(w[kb & 1][nt][st], b[t][st], acc[t][nt]);
  }
}

struct RaggedSmallFwdArgs {
  RaggedFwdArgs r;                             // W[] unused: the weights come from pk
  const float* pk;                             // fragment-major forward weights of the ONE shared slot (k_pack_weights, S = 1)
};

template <int F>
__global__ __launch_bounds__(RGS_THREADS, 3) void k_gnn_fwd_ragged_s(RaggedSmallFwdArgs args) {
  using Lds = RaggedSmallLds<F>;
  using P = FzPack<F>;
  constexpr int FB = Lds::FB, LDT = Lds::LDT, KB = 2 * FB + 1;
  const RaggedFwdArgs& a = args.r;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem + Lds::TILE; float* sS = smem + Lds::SUMS;
  unsigned* sMask = reinterpret_cast<unsigned*>(smem + Lds::MASK);
  int* sInfo = reinterpret_cast<int*>(smem + Lds::INFO);                  // per row: r0 | n << 9 | slot << 17  (slot = RG_SLOTS: none)
  int* sGoff = reinterpret_cast<int*>(smem + Lds::GOFF);
  unsigned* sRec = reinterpret_cast<unsigned*>(smem + Lds::REC);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  typedef const __attribute__((address_space(4))) unsigned char* CBytes;
  const CBytes kargs = (CBytes)__builtin_amdgcn_kernarg_segment_ptr();
  auto stage_ptr = [&](size_t field_off, int s) -> float* {
    return *reinterpret_cast<float* const __attribute__((address_space(4)))*>(kargs + field_off + 8 * (size_t)s);
  };
  const int g0 = a.plan[blockIdx.x], g1 = a.plan[blockIdx.x + 1];
  if (g1 <= g0) return;
  const int R0 = a.graph_off[g0], rows = a.graph_off[g1] - R0, ng = g1 - g0;
  if (rows > RGS_CAP || ng > RGS_CAP) { if (tid == 0 && a.err) atomicOr(a.err, 1); return; }

  // ---- prologue: graph bounds, masks, per-row records, this lane's [x | e] fragments
  for (int i = tid; i <= ng; i += RGS_THREADS) sGoff[i] = a.graph_off[g0 + i] - R0;
  for (int i = tid; i < rows * a.mask_words; i += RGS_THREADS) {
    const int r = i / a.mask_words, w = i - r * a.mask_words;
    sMask[r * RG_MW + w] = a.adjT[(int64_t)(R0 + r) * a.mask_words + w];
  }
  int rr[RGS_RT];
  f32x4 xef[RGS_RT];
#pragma unroll
  for (int t = 0; t < RGS_RT; ++t) {
    rr[t] = 16 * (wv + RGS_WAVES * t) + j;
    xef[t] = ld4(a.xe + (int64_t)(R0 + min(rr[t], rows - 1)) * XE + 4 * kg);
  }
  __syncthreads();                                                        // sGoff, masks visible
  if (tid < rows) {                                                       // a thread per row: its graph by bisection of the tile's offsets
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
    sRec[tid] = rg_row_record(sMask + tid * RG_MW, min(max(n, 0), 128), slot);
  }
  __syncthreads();

  int n_my = 0;                                                           // this wave's row tiles (tiles wv, wv + 4, wv + 8 that hold rows)
#pragma unroll
  for (int t = 0; t < RGS_RT; ++t) n_my += 16 * (wv + RGS_WAVES * t) < rows ? 1 : 0;
  f32x4 agg[RGS_RT][FB];
#pragma unroll
  for (int t = 0; t < RGS_RT; ++t)
#pragma unroll
    for (int b = 0; b < FB; ++b) agg[t][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int s = 0; s <= a.L; ++s) {
    const float* item = s == 0 ? args.pk : args.pk + P::FWD0 + (int64_t)(s - 1) * P::FWD;
    const float* bias = item + (s == 0 ? FB * 256 : KB * FB * 256);
    f32x4 acc[RGS_RT][FB];
#pragma unroll
    for (int t = 0; t < RGS_RT; ++t)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s == 0) {
      if (n_my >= 3) rgs_stage_mfma<F, 3, true>(item, lane, kg, sT, rr, xef, agg, acc);
      else if (n_my == 2) rgs_stage_mfma<F, 2, true>(item, lane, kg, sT, rr, xef, agg, acc);
      else if (n_my == 1) rgs_stage_mfma<F, 1, true>(item, lane, kg, sT, rr, xef, agg, acc);
    } else {
      if (n_my >= 3) rgs_stage_mfma<F, 3, false>(item, lane, kg, sT, rr, xef, agg, acc);
      else if (n_my == 2) rgs_stage_mfma<F, 2, false>(item, lane, kg, sT, rr, xef, agg, acc);
      else if (n_my == 1) rgs_stage_mfma<F, 1, false>(item, lane, kg, sT, rr, xef, agg, acc);
    }
    float* h_out = stage_ptr(offsetof(RaggedFwdArgs, h), s);
    float* a_out = stage_ptr(offsetof(RaggedFwdArgs, a), s);
    f32x4 bv[FB];
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) bv[nt] = ldg4(bias + nt * 16 + 4 * kg);
#pragma unroll
    for (int t = 0; t < RGS_RT; ++t)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) {
        f32x4 v = acc[t][nt] + bv[nt];
        if (s < a.L) v = relu4(v);
        acc[t][nt] = v;
        if (rr[t] < rows) st4(h_out + (int64_t)(R0 + rr[t]) * F + nt * 16 + 4 * kg, v);
      }
    __syncthreads();                                                      // everybody is done with h_{s-1}
#pragma unroll
    for (int t = 0; t < RGS_RT; ++t)
      if (rr[t] < RGS_CAP) {
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) *reinterpret_cast<f32x4*>(sT + rr[t] * LDT + 16 * nt + 4 * kg) = acc[t][nt];
      }
    __syncthreads();                                                      // tile = h_s
    // ---- column sums of the big graphs: slot k by wave k mod 4
    {
      constexpr int COMB = 4 * FB, RGN = 64 / COMB;
      const int cb = lane % COMB, rg = lane / COMB, ckg = cb / FB, ckb = cb - ckg * FB;
      int slot = 0;
      for (int g = 0; g < ng; ++g) {
        const int r0 = sGoff[g], n = sGoff[g + 1] - r0;
        if (n < RG_BIG) continue;
        if ((slot & (RGS_WAVES - 1)) == wv) {
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
    // ---- a_s of this lane's rows (the large-tile kernel's walk, word for word)
#pragma unroll
    for (int t = 0; t < RGS_RT; ++t) {
      f32x4 acc2[FB];
      const bool live = rr[t] < rows;
      const int row = min(rr[t], RGS_CAP - 1);
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
      if (rec & RG_REC_MORE) {
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
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, small tiles: the large-tile kernel (k_gnn_bwd_ragged) with the transposed weights as fragments from L2 (pk_bwd:
// [k-block][2 FB n-tiles][64 lanes][4]: n-tiles 0 .. FB - 1 = dh columns, FB .. 2 FB - 1 = dagg columns)
// ------------------------------------------------------------------------------------------------------------------------
struct RaggedSmallBwdArgs {
  RaggedBwdArgs r;
  const float* pk;                             // fragment-major backward weights of the shared slot
};

template <int F>
struct RaggedSmallBwdLds {
  static constexpr int FB = F / 16, LDT = F + 4;
  static constexpr int TILE = 0, SUMS = TILE + RGS_CAP * LDT, MASK = SUMS + RGS_SLOTS * LDT, INFO = MASK + RGS_CAP * RG_MW,
                       GOFF = INFO + RGS_CAP, TOTAL = GOFF + RGS_CAP + 8;
};

template <int F>
__global__ __launch_bounds__(RGS_THREADS, 2) void k_gnn_bwd_ragged_s(RaggedSmallBwdArgs args) {
  using Lds = RaggedSmallBwdLds<F>;
  using P = FzPack<F>;
  constexpr int FB = Lds::FB, LDT = Lds::LDT;
  const RaggedBwdArgs& a = args.r;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem + Lds::TILE; float* sS = smem + Lds::SUMS;
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
  if (rows > RGS_CAP || ng > RGS_CAP) { if (tid == 0 && a.err) atomicOr(a.err, 1); return; }

  for (int i = tid; i <= ng; i += RGS_THREADS) sGoff[i] = a.graph_off[g0 + i] - R0;
  for (int i = tid; i < rows * a.mask_words; i += RGS_THREADS) {
    const int r = i / a.mask_words, w = i - r * a.mask_words;
    sMask[r * RG_MW + w] = a.adj[(int64_t)(R0 + r) * a.mask_words + w];
  }
  int rr[RGS_RT];
  f32x4 dh[RGS_RT][FB];
#pragma unroll
  for (int t = 0; t < RGS_RT; ++t) {
    rr[t] = 16 * (wv + RGS_WAVES * t) + j;
    const float* g = a.gha + (int64_t)(R0 + min(rr[t], rows - 1)) * (2 * F) + 4 * kg;
#pragma unroll
    for (int b = 0; b < FB; ++b) dh[t][b] = ld4(g + 16 * b);
    if (rr[t] < rows) {
#pragma unroll
      for (int b = 0; b < FB; ++b) *reinterpret_cast<f32x4*>(sT + rr[t] * LDT + 16 * b + 4 * kg) = ld4(g + F + 16 * b);
    }
  }
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
  __syncthreads();                                                        // tile = dagg_L, records, masks

#pragma unroll 1
  for (int s = a.L; s >= 0; --s) {
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
        if ((slot & (RGS_WAVES - 1)) == wv) {
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
    const float* item = args.pk + (int64_t)(max(s, 1) - 1) * P::BWD;
    gvec_p wp = (gvec_p)item + lane;
    f32x4 dagg_new[RGS_RT][FB];
#pragma unroll
    for (int t = 0; t < RGS_RT; ++t) {
      f32x4 gate[FB];
#pragma unroll
      for (int b = 0; b < FB; ++b) gate[b] = gate_next[b];
      if (s < a.L && t + 1 < RGS_RT) load_gate(t + 1, gate_next);
      f32x4 dpre[FB];
#pragma unroll
      for (int b = 0; b < FB; ++b) dpre[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const bool mine = 16 * (wv + RGS_WAVES * t) < rows;                 // (wave-uniform) this row tile holds rows
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
      // ---- [dh_{s-1} | dagg_{s-1}] of the tile = dpre_s . Wt_s, fragments one k-block ahead
      f32x4 acc[2 * FB];
#pragma unroll
      for (int nt = 0; nt < 2 * FB; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (mine) {
        f32x4 w[2][2 * FB];
#pragma unroll
        for (int nt = 0; nt < 2 * FB; ++nt) w[0][nt] = wp[nt * 64];
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) {
          if (kb + 1 < FB) {
#pragma unroll
            for (int nt = 0; nt < 2 * FB; ++nt) w[(kb + 1) & 1][nt] = wp[((kb + 1) * 2 * FB + nt) * 64];
          }
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int nt = 0; nt < 2 * FB; ++nt) acc[nt] = V2X_MFMA(w[kb & 1][nt][st], dpre[kb][st], acc[nt]);
        }
      }
#pragma unroll
      for (int b = 0; b < FB; ++b) { dh[t][b] = acc[b]; dagg_new[t][b] = acc[FB + b]; }
    }
    if (s == 0) break;
    __syncthreads();                                                      // everybody is done with dagg_s
#pragma unroll
    for (int t = 0; t < RGS_RT; ++t)
      if (rr[t] < RGS_CAP) {
#pragma unroll
        for (int b = 0; b < FB; ++b) *reinterpret_cast<f32x4*>(sT + rr[t] * LDT + 16 * b + 4 * kg) = dagg_new[t][b];
      }
    __syncthreads();                                                      // tile = dagg_{s-1}
  }
}

}  // namespace v2x
