// Wide-feature (feat_dim >= 128) node-update kernels: BASELINE config 4 (100 links, feat_dim 256, 3 layers).
//
// At F = 64 a slot's whole weight matrix fits in LDS next to register-resident activation fragments
// (kernels.hpp).  At F = 256 one GNN stage is a [528 x 256] matrix (540 KB) and a node row has 33 K-blocks, so
// the contraction is tiled classically instead: both operands go through LDS in 16-deep K chunks (double
// buffered, one barrier per chunk) and a workgroup owns a 128-row x 64-column output tile of ONE slot.
// This regime is fp32-MFMA bound (45 flop/B at F = 256), 32 MFMAs per wave and chunk against
// 18 LDS reads.  Same "swapped" operand convention as kernels.hpp: weights in A, activations in B, a lane ends
// up with 4 consecutive output features of one node row.
//
//   k_wide_gemm<false>  GNNLayer.call (BS_brain.py:44-51) / Dense-0 (:176):  out = act([seg0|seg1|seg2] W + b)
//   k_wide_gemm<true>   their data gradients:  [dh | dagg] = dpre . W[h rows | agg rows]^T
//   k_wide_wgrad        dW = in^T . dpre, db = sum dpre   (64 x 64 tile of dW per workgroup, rows streamed)
#pragma once
#include <type_traits>
#include "kernels.hpp"

namespace v2x {

struct WideSeg { const float* ptr; int stride; int width; };   // width = padded K columns, multiple of 16

struct WideGemmArgs {
  WideSeg seg[3]; int n_seg;            // the activation row is the concatenation of the segments
  const float* W; int64_t slot_stride;  // layer base (slot 0) in the flat parameter buffer
  RowPad pad;                           // forward: padded K row -> real weight row
  int n_real;                           // weight row length (the layer's output width)
  int k_total;                          // contraction length, multiple of 16
  int n_out;                            // output columns (forward: n_real; transposed: 2F)
  int split, skip;                      // transposed: output column n reads weight row  n < split ? n : n + skip
  float* out; int out_stride;
  int relu, has_bias;
  int n_idx, row_stride, base_mul, idx_base;
};

constexpr int WD_TM = 128, WD_KC = 16;

// NT = output tile width in 16-column MFMA tiles (4: 64 columns; 5: Dense-0's 80 outputs as one strip); the workgroup tile is
// 128 rows x 16 NT columns, K in chunks of 16 through a double-buffered LDS tile.  Measured at configs[3]'s share (profiles/HISTORY.md
// 3b): 64-column tiles at five workgroups per CU beat whole-width tiles, 256-row tiles and 32-deep chunks (L2 absorbs the
// operand re-reads, occupancy matters more).
//
// Round 4 -- the loop's address arithmetic.  Taking things out of the kernel (results then wrong) showed what the launch
// pays for: without ANY LDS read 252 us (of 256), without the barrier 247, without the global loads and their LDS stores
// 208 -- and a load requested two chunks ahead instead of one changes nothing.  It is not latency: the ~60 VALU / SALU
// instructions per chunk that recomputed, for every load, the segment of the K column, the padded -> real weight row, a
// 64-bit row x stride product and a 0 / 1 mask multiply took issue cycles the MFMAs of the SIMD's other waves could not use
// (5 waves x 60 x 4 cycles against 5 x 1024 cycles of MFMA per chunk round = the 18 % that were missing).  Now every thread
// keeps running pointers: an activation pointer per row pass that advances by 16 floats per chunk and jumps by a precomputed
// per-thread distance at the (uniform) chunk where the next input segment starts, and one weight pointer that advances by
// 16 rows, the pad rows of the [x | e] block read from a clamped address and zeroed (one compare + two packed multiplies);
// rows past the slot's last are clamped duplicates whose results are never stored.
template <bool TRANS, int NT, int RT>
__device__ __forceinline__ void wide_gemm_body(const WideGemmArgs& a, float* sAp, float* sBp, const int slot, const int m0, const int n0) {
  constexpr int KC = WD_KC;
  constexpr int TM = 64 * RT, LDB = KC + 4;
  constexpr int TN = 16 * NT, LDA = TN + 4;
  constexpr int PA = (KC * NT + 63) / 64;                        // float4 passes of the weight chunk [KC][TN]
  constexpr int NA = KC * NT * 4;                                // float4 elements of the weight chunk
  constexpr int PB = TM * KC / 1024;                             // float4 passes of the activation chunk [TM][KC]
  constexpr int CPR = KC / 4;                                    // float4 per activation row of the chunk
  float (*sA)[KC * LDA] = reinterpret_cast<float (*)[KC * LDA]>(sAp);    // weights  [2][k][n]
  float (*sB)[TM * LDB] = reinterpret_cast<float (*)[TM * LDB]>(sBp);    // activations [2][row][k]
  const float* Wg = a.W + slot * a.slot_stride;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  const int n_chunks = a.k_total / KC;

  // ---- running pointers (explicit GLOBAL address space: a pointer that is advanced by a per-thread distance between two
  // kernel arguments would otherwise decay to a generic one -- FLAT loads, which the wait-count pass cannot count).
  // Activations: pass q covers tile row (tid + 256 q) / 4, k-columns 4 (tid % 4) .. + 3 of the chunk
  static_assert(PB == RT && RT <= 2 && PA <= 2, "staging registers are named explicitly (arrays of them ended up in scratch)");
  typedef const __attribute__((address_space(1))) f32x4* gf4_p;          // (ext-vector type: HIP's float4 class has no address-space-qualified copy)
  const int cB = (tid % CPR) << 2;
  const int c1 = a.seg[0].width / KC, c2 = a.n_seg > 1 ? c1 + a.seg[1].width / KC : 1 << 30;   // first chunk of segment 1 / 2
  const int64_t row0 = (int64_t)(a.idx_base + min(m0 + tid / CPR, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
  const int64_t row1 = RT > 1 ? (int64_t)(a.idx_base + min(m0 + (tid + 256) / CPR, a.n_idx - 1)) * a.row_stride + slot * a.base_mul : row0;
  gfloat_p pB0 = (gfloat_p)a.seg[0].ptr + row0 * a.seg[0].stride + cB, pB1 = (gfloat_p)a.seg[0].ptr + row1 * a.seg[0].stride + cB;
  // Weights.  Forward: pass q holds k row ka of the chunk and 4 output columns; its real weight row is the padded row minus
  // the pad rows below it, pad rows themselves clamped to the last real row before them.  Transposed (data gradients): output
  // column nT = weight row (skipping `skip` rows at `split`), 4 consecutive k = 4 consecutive floats of that row.
  gfloat_p pA0, pA1;
  int kaA0 = 0, kaA1 = 0;
  {
    const int i0 = min(tid, NA - 1), i1 = min(tid + 256, NA - 1);
    if (!TRANS) {
      kaA0 = i0 / (4 * NT); kaA1 = i1 / (4 * NT);
      pA0 = (gfloat_p)Wg + min(n0 + ((i0 % (4 * NT)) << 2), a.n_real - 4);
      pA1 = (gfloat_p)Wg + min(n0 + ((i1 % (4 * NT)) << 2), a.n_real - 4);
    } else {
      const int nT0 = min(n0 + i0 / CPR, a.n_out - 1), nT1 = min(n0 + i1 / CPR, a.n_out - 1);
      pA0 = (gfloat_p)Wg + (int64_t)(nT0 < a.split ? nT0 : nT0 + a.skip) * a.n_real + ((i0 % CPR) << 2);
      pA1 = (gfloat_p)Wg + (int64_t)(nT1 < a.split ? nT1 : nT1 + a.skip) * a.n_real + ((i1 % CPR) << 2);
    }
  }
  const int pad_at = a.pad.pad_at, pad_end = a.pad.pad_at + a.pad.n_pad, n_pad = a.pad.n_pad, k_last = a.pad.k_real - 1;

  int kc_next = 0;                                               // the chunk the pointers point at
  // one set of staging registers, loads one chunk ahead (two chunks ahead with a second set measured the same: 256 / 237 us)
  f32x4 va0_0, va1_0 = (f32x4){0.f, 0.f, 0.f, 0.f}, vb0_0, vb1_0 = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto load_a = [&](gfloat_p p, int ka) -> f32x4 {
    if (!TRANS) {
      const int kk = kc_next * KC + ka;
      const int rr = min(kk < pad_at ? kk : (kk < pad_end ? pad_at - 1 : kk - n_pad), k_last);
      const f32x4 t = *reinterpret_cast<gf4_p>(p + (int64_t)rr * a.n_real);
      // pad rows are ZERO weights: Dense-0 reads the packed [x | e] block but has no weights for e (its pad rows meet
      // non-zero data); the address stays valid (clamped), the value is dropped
      const float mk = (kk >= pad_at && kk < pad_end) ? 0.f : 1.f;
      return t * mk;
    }
    return *reinterpret_cast<gf4_p>(p + kc_next * KC);
  };
#define V2X_WG_GLOAD(S)                                                                                            \
  {                                                                                                                \
    vb0_##S = *reinterpret_cast<gf4_p>(pB0);                                                                       \
    if (RT > 1) vb1_##S = *reinterpret_cast<gf4_p>(pB1);                                                           \
    va0_##S = load_a(pA0, kaA0);                                                                                   \
    if (PA > 1) va1_##S = load_a(pA1, kaA1);                                                                       \
    if (kc_next + 1 < n_chunks) { /* else: the last prefetch re-loads the last chunk into the idle buffer */       \
      ++kc_next;                                                                                                   \
      pB0 += KC; pB1 += KC;                                                                                        \
      if (kc_next == c1) { /* uniform; twice per workgroup: the next input segment's rows */                       \
        pB0 = (gfloat_p)a.seg[1].ptr + row0 * a.seg[1].stride + cB; pB1 = (gfloat_p)a.seg[1].ptr + row1 * a.seg[1].stride + cB; \
      }                                                                                                            \
      if (kc_next == c2) {                                                                                         \
        pB0 = (gfloat_p)a.seg[2].ptr + row0 * a.seg[2].stride + cB; pB1 = (gfloat_p)a.seg[2].ptr + row1 * a.seg[2].stride + cB; \
      }                                                                                                            \
    }                                                                                                              \
  }
  auto store_a = [&](int buf, int i, f32x4 v) {
    if (i >= NA) return;
    if (!TRANS) {
      *reinterpret_cast<f32x4*>(&sA[buf][(i / (4 * NT)) * LDA + ((i % (4 * NT)) << 2)]) = v;
    } else {
      const int na = i / CPR, ka = (i % CPR) << 2;
      sA[buf][(ka + 0) * LDA + na] = v[0]; sA[buf][(ka + 1) * LDA + na] = v[1];
      sA[buf][(ka + 2) * LDA + na] = v[2]; sA[buf][(ka + 3) * LDA + na] = v[3];
    }
  };
#define V2X_WG_LSTORE(buf, S)                                                                                      \
  {                                                                                                                \
    *reinterpret_cast<f32x4*>(&sB[buf][(tid / CPR) * LDB + cB]) = vb0_##S;                                         \
    if (RT > 1) *reinterpret_cast<f32x4*>(&sB[buf][((tid + 256) / CPR) * LDB + cB]) = vb1_##S;                     \
    store_a(buf, tid, va0_##S);                                                                                    \
    if (PA > 1) store_a(buf, tid + 256, va1_##S);                                                                  \
  }

  f32x4 acc[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define V2X_WG_MFMA(buf)                                                                                           \
  {                                                                                                                \
    f32x4 b[RT];                                                                                                   \
    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) b[rt] = ld4(&sB[buf][(16 * RT * wv + 16 * rt + j) * LDB + 4 * kg]); \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                               \
      float w[NT];                                                                                                 \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) w[nt] = sA[buf][(4 * kg + s) * LDA + nt * 16 + j];        \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                           \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) acc[rt][nt] = V2X_MFMA(w[nt], b[rt][s], acc[rt][nt]);   \
    }                                                                                                              \
  }

  V2X_WG_GLOAD(0)
  V2X_WG_LSTORE(0, 0)
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc < n_chunks; ++kc) {
    const int buf = kc & 1;
    // unconditional prefetch (a load under `if` makes hipcc wait for it at the join, i.e. BEFORE the MFMAs it is supposed
    // to overlap); the last iteration re-loads its own chunk into the idle buffer
    V2X_WG_GLOAD(0)
    __builtin_amdgcn_sched_barrier(0);
    V2X_WG_MFMA(buf)
    __builtin_amdgcn_sched_barrier(0);
    V2X_WG_LSTORE(buf ^ 1, 0)
    __syncthreads();
  }

#undef V2X_WG_MFMA
#undef V2X_WG_GLOAD
#undef V2X_WG_LSTORE
  // epilogue: lane holds out[row 16*RT*wv + 16*rt + j][n0 + nt*16 + 4*kg .. +3]
  const float* bias = Wg + (int64_t)a.pad.k_real * a.n_real;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int idx = m0 + 16 * RT * wv + 16 * rt + j;
    if (idx >= a.n_idx) continue;
    const int64_t rowo = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + nt * 16 + 4 * kg;
      if (col >= a.n_out) continue;
      f32x4 v = acc[rt][nt];
      if (!TRANS && a.has_bias) v = v + ld4(bias + col);
      if (!TRANS && a.relu) v = relu4(v);
      st4(a.out + rowo * a.out_stride + col, v);
    }
  }
}


// Tail of the launch as HALF tiles.  The node update of configs[3]'s share is 3,200 tiles of 128 x 64 on 1,280 resident
// workgroups (five per CU): 2.5 rounds, i.e. the third round runs half empty and the launch lasts three tile times (VERDICT
// r04: up to 17 % of it).  Row blocks [0, n_full_rb) of the global enumeration (slot-major) run as 128-row tiles; the
// remaining ones -- the fraction beyond the last full round, chosen by the host -- as two 64-row tiles each, dispatched LAST
// (1-D grid: the dispatcher hands out workgroups in order), so that the round that was half empty is one full round of
// half-size tiles.  Same arithmetic per output element (a tile's rows are independent): bitwise the same results.
// Workgroup -> tile, XCD-aware, in both regions: workgroup ids are dealt round-robin to the 8 XCDs, each with an L2 of its own;
// the ids i, i + 8, i + 16, ... of a run of 8 ny consecutive workgroups are the ny column tiles of ONE row block (or half),
// so its activation strip ([TM x K]: 270 KB at K = 528) is fetched from HBM once and hit in that L2 by the others (with the
// row block as the fastest index the strip came from HBM once per column tile: round 4, profiles/HISTORY.md 3b).  The last, partial
// run keeps the plain order.
struct WideTiling { int ny, mbs, n_full_rb, n_rb; };
__device__ __forceinline__ void wide_run_order(int lin, int n_units, int ny, int& unit, int& nb) {
  const int run = 8 * ny, full = n_units / 8 * run;
  if (lin < full) { unit = lin / run * 8 + (lin & 7); nb = (lin % run) >> 3; }
  else { const int l = lin - full; unit = full / ny + l / ny; nb = l % ny; }
}

template <bool TRANS, int NT>
__global__ __launch_bounds__(256, 5) void k_wide_gemm(WideGemmArgs a, WideTiling t) {      // five workgroups per CU (29 / 31 KB of LDS each): <= 96 registers
  constexpr int KC = WD_KC, LDB = KC + 4, LDA = 16 * NT + 4;
  __shared__ __attribute__((aligned(16))) float sA[2 * KC * LDA];
  __shared__ __attribute__((aligned(16))) float sB[2 * 128 * LDB];
  const int lin = blockIdx.x, n_full = t.n_full_rb * t.ny;
  int rb, nb, half = -1;
  if (lin < n_full) {
    wide_run_order(lin, t.n_full_rb, t.ny, rb, nb);
  } else {
    int rh;
    wide_run_order(lin - n_full, 2 * (t.n_rb - t.n_full_rb), t.ny, rh, nb);
    rb = t.n_full_rb + (rh >> 1); half = rh & 1;
  }
  const int slot = rb / t.mbs, mb = rb - slot * t.mbs, n0 = nb * 16 * NT;
  if (half < 0) {
    wide_gemm_body<TRANS, NT, 2>(a, sA, sB, slot, mb * 128, n0);
  } else {
    const int m0 = mb * 128 + 64 * half;
    if (m0 < a.n_idx) wide_gemm_body<TRANS, NT, 1>(a, sA, sB, slot, m0, n0);
  }
}

// ------------------------------------------------------------------------------------------------------------
struct WideWgradArgs {
  WideSeg seg[3]; int n_seg;             // K operand; a 64-wide K tile never straddles two segments
  int seg_kpad[3];                       // padded K row where each segment starts
  // optional 16-wide segment (the packed [x | e] rows) that rides on the workgroup of K tile 0 instead of owning a K tile:
  // as a tile of its own it cost a full KW x TN tile's MFMAs for 16 / KW useful rows -- one fifth of a GNN stage's launch
  // at F = 256 (round 3 tried to skip its zero strips inside the loop: the branch cost more than it saved)
  WideSeg xseg; int xseg_kpad;
  const float* dpre; int d_stride; int n_real;
  RowPad pad;
  float* slab; int64_t slab_stride;      // slab[split][P]
  int64_t layer_off, slot_stride;
  int n_idx, row_stride, base_mul, idx_base;
  int n_split, rows_per_split;           // rows_per_split is a multiple of 32
  // single-GPU fit step, one row split: the workgroup holds the layer's FINAL gradient tile -- it applies Keras Adam to its
  // parameters right there (param / mom / vel: the flat buffers; adam_scal = {lr_t, beta1, beta2, eps} in device memory, so
  // that a replayed hipGraph sees this step's lr_t) and the gradient never travels: - 8 of the 28 bytes per parameter that
  // k_reduce_adam moved, and the other 20 ride under the launch's MFMA work instead of being a 280 us launch of their own
  float* param; float* mom; float* vel; const float* adam_scal; int adam;
};


// KW = K-tile width (input features per workgroup: 64 or 128, a wave owns KW/64 strips of 16), NT = output tile width
// in 16-column tiles (5: Dense-0, 8: F = 128, 16: 256 columns).  Whole-width output tiles for the same reason as in
// k_wide_gemm: with 64 x 64 tiles every input tile was fetched once per 64 output columns and every dpre tile once
// per 64 input features -- 1.9 GB per launch at 100 links x 1024 graphs x 256 features.
// FOLD: this workgroup also accumulates the 16 rows of a.xseg against the dpre tile it stages anyway; wave w owns output
// tiles w, w + 4, ... of that strip (+ 12.5 % MFMAs for one workgroup in KT instead of a whole extra workgroup).
template <int KW, int NT> struct WideWgradLds {
  static constexpr int WW_TR = NT >= 16 ? 16 : 32;               // rows per chunk (LDS budget: 2 workgroups per CU)
  // row strides = 16 mod 64 banks: the operand reads are dwords at [row 4 s + kg][16-aligned offset + j] -- the four k-groups
  // of a wave land on four disjoint sets of 16 banks (with + 4, round 3, they overlapped: up to 4 lanes per bank)
  static constexpr int LDX = KW + 16, LDD = 16 * NT + ((16 * NT) % 64 == 0 ? 16 : (16 * NT) % 64 == 16 ? 0 : (16 * NT) % 64 == 32 ? 48 : 32), LDE = 16;
  static constexpr int X = 2 * WW_TR * LDX, D = 2 * WW_TR * LDD, E = 2 * WW_TR * LDE;
};

template <int KW, int NT, bool FOLD, bool ADAM = false>
__device__ __forceinline__ void wide_wgrad_body(const WideWgradArgs& a, float* sXp, float* sDp, float* sEp, int bx, int by, int bz) {
  typedef WideWgradLds<KW, NT> Lds;
  constexpr int WW_TR = Lds::WW_TR;
  constexpr int KS = KW / 64, LDX = Lds::LDX, XP = KW * WW_TR / 1024;   // strips per wave, LDS stride, float4 passes of the X tile
  constexpr int TN = 16 * NT, LDD = Lds::LDD, DP = (WW_TR * TN / 4 + 255) / 256;   // dpre tile [WW_TR][TN]
  constexpr int LDE = Lds::LDE, NE = (NT + 3) / 4;              // folded strip: [WW_TR][16] tile, output tiles per wave
  float (*sX)[WW_TR * LDX] = reinterpret_cast<float (*)[WW_TR * LDX]>(sXp);
  float (*sD)[WW_TR * LDD] = reinterpret_cast<float (*)[WW_TR * LDD]>(sDp);
  float (*sE)[WW_TR * LDE] = reinterpret_cast<float (*)[WW_TR * LDE]>(sEp);
  const int slot = bz / a.n_split, sp = bz - slot * a.n_split;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  // K tile -> (segment, column offset, valid width)
  int kt = bx;
  const float* xp = a.seg[0].ptr; int xst = a.seg[0].stride, xw = a.seg[0].width, kpad0 = a.seg_kpad[0];
  {
    const int t0 = (a.seg[0].width + KW - 1) / KW;
    if (a.n_seg > 1 && kt >= t0) {
      kt -= t0; xp = a.seg[1].ptr; xst = a.seg[1].stride; xw = a.seg[1].width; kpad0 = a.seg_kpad[1];
      const int t1 = (a.seg[1].width + KW - 1) / KW;
      if (a.n_seg > 2 && kt >= t1) { kt -= t1; xp = a.seg[2].ptr; xst = a.seg[2].stride; xw = a.seg[2].width; kpad0 = a.seg_kpad[2]; }
    }
  }
  const int kcol0 = kt * KW, kw = min(KW, xw - kcol0);
  kpad0 += kcol0;
  const int n0 = by * TN;
  const int i_begin = sp * a.rows_per_split, i_end = min(i_begin + a.rows_per_split, a.n_idx);
  const int n_chunks = (max(i_end - i_begin, 0) + WW_TR - 1) / WW_TR;

  auto gload = [&](int c, float4 (&vx)[XP], float4 (&vd)[DP], float4& ve) {
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int i = tid + 256 * p, r = i / (KW / 4), cx = (i % (KW / 4)) << 2;
      const int idx = i_begin + c * WW_TR + r;
      const bool okx = cx < kw;
      const int64_t rowg = (int64_t)(a.idx_base + min(idx, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
      const float4 x = *reinterpret_cast<const float4*>(xp + rowg * xst + kcol0 + (okx ? cx : 0));
      const float mx = okx ? 1.f : 0.f;              // (rows past the split are zeroed on the dpre side)
      vx[p] = make_float4(x.x * mx, x.y * mx, x.z * mx, x.w * mx);
    }
#pragma unroll
    for (int p = 0; p < DP; ++p) {
      const int i = tid + 256 * p, r = min(i / (TN / 4), WW_TR - 1), cd = n0 + ((i % (TN / 4)) << 2);
      const int idx = i_begin + c * WW_TR + r;
      const bool okd = i < WW_TR * TN / 4 && cd < a.n_real && idx < i_end;
      const int64_t rowg = (int64_t)(a.idx_base + min(idx, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
      const float4 d = *reinterpret_cast<const float4*>(a.dpre + rowg * a.d_stride + (cd < a.n_real ? cd : 0));
      const float md = okd ? 1.f : 0.f;
      vd[p] = make_float4(d.x * md, d.y * md, d.z * md, d.w * md);
    }
    if (FOLD) {                                       // the [x | e] rows of the chunk: thread (row tid / 4, 4 columns); others repeat
      const int r = min(tid >> 2, WW_TR - 1), idx = i_begin + c * WW_TR + r;
      const int64_t rowg = (int64_t)(a.idx_base + min(idx, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
      ve = *reinterpret_cast<const float4*>(a.xseg.ptr + rowg * a.xseg.stride + ((tid & 3) << 2));
    }
  };
  auto lstore = [&](int buf, const float4 (&vx)[XP], const float4 (&vd)[DP], const float4& ve) {
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int i = tid + 256 * p, r = i / (KW / 4), cx = (i % (KW / 4)) << 2;
      *reinterpret_cast<float4*>(&sX[buf][r * LDX + cx]) = vx[p];
    }
#pragma unroll
    for (int p = 0; p < DP; ++p) {
      const int i = tid + 256 * p;
      if (i < WW_TR * TN / 4) *reinterpret_cast<float4*>(&sD[buf][(i / (TN / 4)) * LDD + ((i % (TN / 4)) << 2)]) = vd[p];
    }
    if (FOLD && tid < 4 * WW_TR) *reinterpret_cast<float4*>(&sE[buf][(tid >> 2) * LDE + ((tid & 3) << 2)]) = ve;
  };

  f32x4 acc[KS][NT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[ks][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accE[NE];
#pragma unroll
  for (int t = 0; t < NE; ++t) accE[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;                                             // column n0 + tid of the kt == 0 tile
  const bool do_bias = bx == 0 && tid < TN;

  // Every chunk but the last of a split is whole and inside the batch: on a full K tile x full column tile its loads need
  // neither clamp nor mask, and go through gload_fast (no vector-ALU address arithmetic at all, where gload spends ~15
  // integer instructions per load on row / segment / mask -- this launch's vector-ALU instructions come out of its MFMA
  // time: a SIMD does not overlap the two, kernels_ragged.hpp); gload keeps the first chunk, the tail and partial tiles.
  typedef const __attribute__((address_space(1))) f32x4* gf4_p;
  // address = UNIFORM chunk base (scalar registers, advanced by scalar adds) + a per-thread 32-bit offset that never changes
  // (the loads' saddr + voffset form): the passes of a tile are a uniform number of rows apart
  constexpr bool REG = 256 % (KW / 4) == 0 && 256 % (TN / 4) == 0 && WW_TR * TN / 4 == 256 * DP;     // (not Dense-0's 80 columns)
  const int64_t row_step = (int64_t)WW_TR * a.row_stride;
  const int64_t row01 = (int64_t)(a.idx_base + i_begin + WW_TR) * a.row_stride + slot * a.base_mul;       // chunk 1 (chunk 0 takes gload)
  const int cx0 = (tid % (KW / 4)) << 2, cd0 = n0 + ((tid % (TN / 4)) << 2);
  const unsigned off_x = (unsigned)((tid / (KW / 4)) * a.row_stride * xst + cx0);
  const unsigned off_d = (unsigned)((tid / (TN / 4)) * a.row_stride * a.d_stride + cd0);
  const unsigned off_e = FOLD ? (unsigned)(min(tid >> 2, WW_TR - 1) * a.row_stride * a.xseg.stride + ((tid & 3) << 2)) : 0u;
  gfloat_p bx_ = (gfloat_p)xp + row01 * xst + kcol0, bd_ = (gfloat_p)a.dpre + row01 * a.d_stride;
  gfloat_p be_ = FOLD ? (gfloat_p)a.xseg.ptr + row01 * a.xseg.stride : nullptr;
  const int64_t pass_x = (int64_t)(256 / (KW / 4)) * a.row_stride * xst, pass_d = (int64_t)(256 / (TN / 4)) * a.row_stride * a.d_stride;
  const int64_t step_x = row_step * xst, step_d = row_step * a.d_stride, step_e = FOLD ? row_step * a.xseg.stride : 0;
  // plain: every thread's loads of a whole chunk are real elements (full K tile, full column tile, whole passes)
  const bool plain = REG && kw == KW && n0 + TN <= a.n_real;
  auto gload_fast = [&](float4 (&vx)[XP], float4 (&vd)[DP], float4& ve) {     // the NEXT whole chunk; the bases move on
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const f32x4 x = *reinterpret_cast<gf4_p>(bx_ + p * pass_x + off_x);
      vx[p] = make_float4(x[0], x[1], x[2], x[3]);
    }
#pragma unroll
    for (int p = 0; p < DP; ++p) {
      const f32x4 d = *reinterpret_cast<gf4_p>(bd_ + p * pass_d + off_d);
      vd[p] = make_float4(d[0], d[1], d[2], d[3]);
    }
    if (FOLD) {
      const f32x4 e = *reinterpret_cast<gf4_p>(be_ + off_e);
      ve = make_float4(e[0], e[1], e[2], e[3]);
      be_ += step_e;
    }
    bx_ += step_x; bd_ += step_d;
  };

  float4 vx[XP], vd[DP], ve = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n_chunks > 0) { gload(0, vx, vd, ve); lstore(0, vx, vd, ve); }
  __syncthreads();
  // (two loops over one body: a branch between the two loaders INSIDE the loop cost 40 us of this launch)
  auto chunk = [&](int c, auto fast) {
    const int buf = c & 1;
    if constexpr (decltype(fast)::value) gload_fast(vx, vd, ve);
    else gload(min(c + 1, n_chunks - 1), vx, vd, ve);  // first / last chunks, partial tiles: clamped and masked; unconditional
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < WW_TR / 4; ++s) {
      float xa[KS], db[NT];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xa[ks] = sX[buf][(4 * s + kg) * LDX + 64 * ks + 16 * wv + j];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) db[nt] = sD[buf][(4 * s + kg) * LDD + nt * 16 + j];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc[ks][nt] = V2X_MFMA(xa[ks], db[nt], acc[ks][nt]);
      if (FOLD) {
        const float xe = sE[buf][(4 * s + kg) * LDE + j];
#pragma unroll
        for (int t = 0; t < NE; ++t) {                // this wave's tiles of the folded strip: operands re-read from LDS
          const float de = sD[buf][(4 * s + kg) * LDD + min(wv + 4 * t, NT - 1) * 16 + j];      // (db[] cannot be indexed by wv)
          accE[t] = V2X_MFMA(xe, de, accE[t]);
        }
      }
    }
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < WW_TR; ++r) bsum += sD[buf][r * LDD + tid];
    }
    __builtin_amdgcn_sched_barrier(0);
    lstore(buf ^ 1, vx, vd, ve);
    __syncthreads();
  };
  const int n_fast = plain ? max(n_chunks - 2, 0) : 0;         // chunk c loads chunk c + 1: whole and in range for c + 1 < n_chunks - 1
#pragma unroll 1
  for (int c = 0; c < n_fast; ++c) chunk(c, std::true_type());
#pragma unroll 1
  for (int c = n_fast; c < n_chunks; ++c) chunk(c, std::false_type());

  // lane holds dW[kpad0 + 64*ks + 16*wv + 4*kg + r][n0 + nt*16 + j]
  const int64_t lbase = a.layer_off + slot * a.slot_stride;
  float* dst = a.slab + (int64_t)sp * a.slab_stride + lbase;
  float lr_t = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f;
  if (ADAM) { lr_t = a.adam_scal[0]; b1 = a.adam_scal[1]; b2 = a.adam_scal[2]; eps = a.adam_scal[3]; }
  float* pp = a.param + lbase; float* pm = a.mom + lbase; float* pv = a.vel + lbase;
  // one final gradient value: to the gradient buffer / slab, or straight through Keras Adam (k_reduce_adam's expressions)
  auto emit = [&](int64_t off, float g) {
    if (ADAM) {
      float mo = pm[off], ve = pv[off], p = pp[off];
      mo = b1 * mo + (1.f - b1) * g;
      ve = b2 * ve + (1.f - b2) * (g * g);
      p -= lr_t * mo / (sqrtf(ve) + eps);
      pm[off] = mo; pv[off] = ve; pp[off] = p;
    } else {
      dst[off] = g;
    }
  };
  // The accumulators leave through a wave-private LDS transpose: for a fixed (strip, r) a wave holds 4 weight rows x TN
  // columns with a lane's values 16 columns apart; written to LDS and read back along the rows, every lane has float4s of
  // consecutive columns -- 16-byte accesses on whole rows instead of 4-byte ones on 64-byte pieces (x 3 loads + 3 stores
  // per element with Adam on board: the scalar form made the merged launch 200 us longer than the Adam launch it replaced;
  // this form 105 us).  After the loop's last barrier nobody reads the operand tiles any more; a wave's LDS accesses
  // execute in order.
  constexpr int TNS = TN + ((TN % 64 == 16 || TN % 64 == 48) ? 0 : 16);     // row stride = 16 mod 32 banks: conflict-free writes
  constexpr int C4 = TN / 4, NP4 = (4 * C4 + 63) / 64;
  static_assert(4 * 4 * TNS <= 2 * WW_TR * LDD, "transpose buffer must fit the dpre tiles");
  float* wbuf = sDp + wv * (4 * TNS);
  auto emit4 = [&](int64_t off, f32x4 g) {
    if (ADAM) {
      f32x4 mo = ld4(pm + off), ve = ld4(pv + off), p = ld4(pp + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mo[e] = b1 * mo[e] + (1.f - b1) * g[e];
        ve[e] = b2 * ve[e] + (1.f - b2) * (g[e] * g[e]);
        p[e] -= lr_t * mo[e] / (sqrtf(ve[e]) + eps);
      }
      st4(pm + off, mo); st4(pv + off, ve); st4(pp + off, p);
    } else {
      st4(dst + off, g);
    }
  };
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wbuf[kg * TNS + nt * 16 + j] = acc[ks][nt][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int p4 = 0; p4 < NP4; ++p4) {
        const int i = lane + 64 * p4, row4 = min(i / C4, 3), c4 = i % C4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(wbuf + row4 * TNS + 4 * c4);
        const int kk = 64 * ks + 16 * wv + 4 * row4 + r;
        const int rr = kk < kw ? real_row(a.pad, kpad0 + kk) : -1;
        const int col = n0 + 4 * c4;
        if (i < 4 * C4 && rr >= 0 && col < a.n_real) emit4((int64_t)rr * a.n_real + col, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  if (FOLD) {                                         // ... and dW[xseg_kpad + 4*kg + r][n0 + (wv + 4 t)*16 + j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = real_row(a.pad, a.xseg_kpad + 4 * kg + r);
      if (rr < 0) continue;
#pragma unroll
      for (int t = 0; t < NE; ++t) {
        const int col = n0 + (wv + 4 * t) * 16 + j;
        if (wv + 4 * t < NT && col < a.n_real) emit((int64_t)rr * a.n_real + col, accE[t][r]);
      }
    }
  }
  if (do_bias && n0 + tid < a.n_real) emit((int64_t)a.pad.k_real * a.n_real + n0 + tid, bsum);
}

template <int KW, int NT>
__global__ __launch_bounds__(256, 2) void k_wide_wgrad(WideWgradArgs a) {
  typedef WideWgradLds<KW, NT> Lds;
  __shared__ __attribute__((aligned(16))) float sX[Lds::X];
  __shared__ __attribute__((aligned(16))) float sD[Lds::D];
  __shared__ __attribute__((aligned(16))) float sE[Lds::E];
  if (a.xseg.ptr && blockIdx.x == 0) wide_wgrad_body<KW, NT, true>(a, sX, sD, sE, blockIdx.x, blockIdx.y, blockIdx.z);
  else wide_wgrad_body<KW, NT, false>(a, sX, sD, sE, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several layers' weight gradients as ONE grid (single-GPU training: nothing waits for a single layer's gradient).  With
// per-node weights a layer is [K tiles x slots] workgroups of a quarter million MFMA cycles each and a launch holds about two
// of them per CU: 500 workgroups on 256 CUs, every launch as long as its fullest CU -- and the launches of L stages, Dense-0
// and the embed layer each pay that rounding on their own.  As roles of one grid (heaviest first; a 1-D grid cut by
// `start`) the chip is handed 5-6 workgroups per CU to balance: configs[3] share 3 x 298 + 114 + 59 us -> see profiles/HISTORY.md 3b.
constexpr int WWM_ROLES = 6;
enum { WWM_128x16 = 0, WWM_128x8 = 1, WWM_128x5 = 2, WWM_64x4 = 3 };
struct WideWgradMulti {
  WideWgradArgs w[WWM_ROLES];
  int start[WWM_ROLES + 1];              // first workgroup of every role (+ the grid size)
  int kt[WWM_ROLES], nt[WWM_ROLES], kind[WWM_ROLES];
  int n_roles;
};
template <int A, int B> struct CMax { static constexpr int v = A > B ? A : B; };

__global__ __launch_bounds__(256, 2) void k_wide_wgrad_multi(WideWgradMulti mu) {
  constexpr int SX = CMax<CMax<WideWgradLds<128, 16>::X, WideWgradLds<128, 8>::X>::v, CMax<WideWgradLds<128, 5>::X, WideWgradLds<64, 4>::X>::v>::v;
  constexpr int SD = CMax<CMax<WideWgradLds<128, 16>::D, WideWgradLds<128, 8>::D>::v, CMax<WideWgradLds<128, 5>::D, WideWgradLds<64, 4>::D>::v>::v;
  constexpr int SE = WideWgradLds<128, 5>::E;
  __shared__ __attribute__((aligned(16))) float sX[SX];
  __shared__ __attribute__((aligned(16))) float sD[SD];
  __shared__ __attribute__((aligned(16))) float sE[SE];
  // role descriptors through the kernarg segment pointer (a by-value array indexed at run time would be copied to scratch)
  typedef const __attribute__((address_space(4))) unsigned* CWords;
  static_assert(sizeof(WideWgradArgs) % 4 == 0, "WideWgradArgs must be dword sized");
  constexpr int NW = sizeof(WideWgradArgs) / 4;
  CWords base = (CWords)__builtin_amdgcn_kernarg_segment_ptr();
  CWords tail = base + WWM_ROLES * NW;                            // start[WWM_ROLES + 1], kt[], nt[], kind[], n_roles
  const int b = blockIdx.x;
  int r = 0;
#pragma unroll
  for (int i = 1; i < WWM_ROLES; ++i) r += (b >= (int)tail[i] && i < (int)tail[WWM_ROLES + 1 + 3 * WWM_ROLES]) ? 1 : 0;
  const int local = b - (int)tail[r];
  const int kt = (int)tail[WWM_ROLES + 1 + r], nt = (int)tail[WWM_ROLES + 1 + WWM_ROLES + r], kind = (int)tail[WWM_ROLES + 1 + 2 * WWM_ROLES + r];
  // The kt K tiles of one (column tile, row split) chunk all read the SAME dpre rows; with the K tile as the fastest index
  // they sat on kt different XCDs and every L2 fetched the rows for itself (PMC: 3.5 GB per launch against 1.3 GB
  // algorithmic at configs[3]).  As in k_wide_gemm: ids i, i + 8, ... of a run of 8 kt workgroups -- one XCD -- are the K
  // tiles of one chunk.
  int bx, chunk;
  {
    const int n_chunks = ((int)tail[r + 1] - (int)tail[r]) / kt, run = 8 * kt, full = n_chunks / 8 * run;
    if (local < full) { bx = (local % run) >> 3; chunk = local / run * 8 + (local & 7); }
    else { const int l = local - full; bx = l % kt; chunk = full / kt + l / kt; }
  }
  const int by = chunk % nt, bz = chunk / nt;
  WideWgradArgs a;
  unsigned* dstw = reinterpret_cast<unsigned*>(&a);
  CWords srcw = base + r * NW;
#pragma unroll
  for (int i = 0; i < NW; ++i) dstw[i] = srcw[i];
  const bool fold = a.xseg.ptr && bx == 0;
#define V2X_WWM_ROLE(KWV, NTV)                                                                                          \
  {                                                                                                                     \
    if (a.adam) { if (fold) wide_wgrad_body<KWV, NTV, true, true>(a, sX, sD, sE, bx, by, bz); else wide_wgrad_body<KWV, NTV, false, true>(a, sX, sD, sE, bx, by, bz); } \
    else { if (fold) wide_wgrad_body<KWV, NTV, true>(a, sX, sD, sE, bx, by, bz); else wide_wgrad_body<KWV, NTV, false>(a, sX, sD, sE, bx, by, bz); } \
  }
  if (kind == WWM_128x16) V2X_WWM_ROLE(128, 16)
  else if (kind == WWM_128x8) V2X_WWM_ROLE(128, 8)
  else if (kind == WWM_128x5) V2X_WWM_ROLE(128, 5)
  else wide_wgrad_body<64, 4, false>(a, sX, sD, sE, bx, by, bz);
#undef V2X_WWM_ROLE
}


// ------------------------------------------------------------------------------------------------------------
// k_agg_dense : AggLayer.call (BS_brain.py:69-76) for LARGE, DENSE graphs (e.g. 100 links, in-degree 98).
//
// The gather form (k_agg) issues one 16-byte LDS read per edge and lane: at in-degree 98 it is LDS-bandwidth
// bound (measured 360 us per launch at 1024 x 100 x 256, 1.06 TB/s of HBM).  Here the graph's feature tile
// is still staged in LDS once (same HBM traffic), but the contraction out[q] = sum_p Adj[p][q] h[p] runs on
// the fp32 MFMA pipe with the adjacency held as per-source BIT masks (N x ceil(N/32) words, built from the
// CSR with integer atomics) and expanded to 0.0/1.0 B-operand values on the fly:
//   A[i = feature][k = p] = h[p][f0+i]   (LDS, consecutive lanes -> consecutive banks)
//   B[k = p][j = q]       = Adj[p][q]    (bit q of mask[p];   transposed:  B[k = q][j = p] = bit q of mask[p])
//   D[i][j]: lane holds out[row j][4 consecutive features] -> float4 epilogue (+add, ReLU' gate) and store.
// 0/1 times h is exact, so the only difference to the gather form is the fp32 summation order.
//
// Round 3 -- per graph, through the complement when that is cheaper.  The reference's interference graph is complete
// minus two (link q hears every link but itself and its own receiver's), so the N x N x F
// product is spent adding 98 rows where  out[q] = S - sum_{p : Adj[p][q] = 0} h[p],  S = the column sum of the graph's
// rows, needs two.  A workgroup knows its graph's edge count from row_ptr: with at most AD_COMPL_PER_ROW non-edges per
// row on average it sums the columns once (fixed order), then walks the ZERO bits of each output row's mask (ascending)
// -- LDS reads of ~4 N rows instead of N^2 / 4 MFMAs, which leaves the launch to its HBM traffic; any other graph takes
// the MFMA product as before.  Valid for every adjacency (rows with fewer edges than non-edges are summed directly: decided
// per row); differs from the other forms by fp32 rounding only.  The
// forward needs the masks by DESTINATION (bit p of adjT[q]), the transpose by source (bit q of adj[p]): k_adj_masks
// writes both.
constexpr int AD_COMPL_PER_ROW = 8;

// Work plan of the ragged fused graph layers (kernels_ragged.hpp): runs of whole graphs of <= cap rows, each as long as it
// can be.  plan[w] = first graph of run w, = n_graphs past the last run (w = 0 .. W).  The chain "the next run starts where
// this one stops fitting" is resolved without walking it: nxt[g] by a bounded binary search over the offsets for all g at
// once, then log2(W) rounds of pointer doubling, run w composing the powers named by the bits of w.  One workgroup; `off`
// may be the offsets in LDS or in global memory, T the table entry type (unsigned short halves the tables: n_graphs < 65535).
template <typename T, int THREADS, typename OffPtr>
__device__ __forceinline__ void rg_plan_tables(OffPtr off, T* cur, T* oth, T* first, int B, int W, int cap, int32_t* plan, int tid) {
  for (int w = tid; w <= W; w += THREADS) first[w] = (T)0;
  // largest t in (g, min(B, g + cap)] with off[t] - off[g] <= cap; a graph that alone exceeds cap still advances by one (the
  // kernels flag it), B is a fixed point.  Four searches side by side, a fixed number of halvings (the range is <= cap wide;
  // a search that has converged keeps its answer): independent LDS round trips in flight instead of one chain after the other.
  const int n_halvings = 32 - __builtin_clz((unsigned)max(cap, 1));
  for (int g0 = tid; g0 <= B; g0 += 4 * THREADS) {
    int lo[4], hi[4], base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = min(g0 + u * THREADS, B);
      lo[u] = min(g + 1, B); hi[u] = min(B, g + cap); base[u] = off[g];
    }
    for (int it = 0; it < n_halvings; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mid = (lo[u] + hi[u] + 1) >> 1;
        const bool fits = off[mid] - base[u] <= cap;
        lo[u] = fits ? mid : lo[u];
        hi[u] = fits ? hi[u] : max(mid - 1, lo[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (g0 + u * THREADS <= B) cur[g0 + u * THREADS] = (T)lo[u];
  }
  __syncthreads();
  for (int b = 0; (1 << b) <= W; ++b) {
    for (int w = tid; w <= W; w += THREADS)
      if ((w >> b) & 1) first[w] = cur[first[w]];
    for (int g0 = tid; g0 <= B; g0 += 4 * THREADS) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = cur[min(g0 + u * THREADS, B)];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = cur[v[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (g0 + u * THREADS <= B) oth[g0 + u * THREADS] = v[u];
    }
    __syncthreads();
    T* t = cur; cur = oth; oth = t;
  }
  for (int w = tid; w <= W; w += THREADS) plan[w] = (int32_t)first[w];
}

struct AggDenseArgs {
  const float* src; int src_stride;
  const float* add; int add_stride;
  const float* mask;                    // optional [R][F] ReLU' gate
  float* out;
  const int32_t* graph_off; const int32_t* row_ptr; const int32_t* col_idx;
  unsigned* adj;                        // [R][mask_words]: bit q of adj[p] = edge p -> q (graph-local q), built by k_adj_masks
  unsigned* adjT;                       // [R][mask_words]: bit p of adjT[q] = edge p -> q (same edges, by destination)
  int g_base, n_graphs, n_nodes, F;
  int n_fg;                             // 64-wide feature groups per graph (one workgroup each)
  int rows_cap, mask_words;             // rows_cap: max nodes rounded up to 16
  int* err;                             // host-visible flag word (bit 0: a graph exceeds rows_cap)
  // k_adj_masks only: the work plan of the ragged fused forward (kernels_ragged.hpp) -- plan[w] = first graph whose first row
  // is >= w * plan_capp, w = 0 .. plan_n (plan[plan_n] = n_graphs); written by one thread per graph, no search
  int32_t* plan; int plan_capp, plan_n;
  // ... or, plan_cap > 0: the PACKED plan (rg_plan_tables) by workgroup 0 of this launch before its own graph's masks, offsets
  // and 16-bit tables in the launch's dynamic LDS -- it runs beside the other mask workgroups
  int plan_cap;
};

// CSR-by-destination -> per-source bit masks, one workgroup per graph (integer atomics in LDS: order-independent).
// Runs ONCE per batch; the 2(L+1) aggregations of a training step then read 4*ceil(N/32) bytes per node instead
// of 4 bytes per edge.
__global__ __launch_bounds__(256) void k_adj_masks(AggDenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* sM = reinterpret_cast<unsigned*>(smem);                         // [rows_cap][mask_words] by source
  unsigned* sD = sM + a.rows_cap * a.mask_words;                            // [rows_cap][mask_words] by destination
  const int tid = threadIdx.x;
  if (a.plan_cap > 0 && blockIdx.x == 0) {                                   // workgroup 0 makes the plan first (ragged batches: graph_off given)
    int* off = reinterpret_cast<int*>(smem);                                  // (an EXTRA workgroup would be the 2049th of a launch whose
    unsigned short* cur = reinterpret_cast<unsigned short*>(off + a.n_graphs + 1);     //  2048 fill the chip: it would start late)
    unsigned short* oth = cur + a.n_graphs + 1;
    unsigned short* first = oth + a.n_graphs + 1;
    for (int g = tid; g <= a.n_graphs; g += 256) off[g] = a.graph_off[a.g_base + g];
    __syncthreads();
    rg_plan_tables<unsigned short, 256>(off, cur, oth, first, a.n_graphs, a.plan_n, a.plan_cap, a.plan, tid);
    __syncthreads();                                                          // the tables' LDS becomes this workgroup's mask scratch
  }
  const int g = a.g_base + blockIdx.x;
  const int r_begin = a.graph_off ? a.graph_off[g] : g * a.n_nodes;
  const int n = (a.graph_off ? a.graph_off[g + 1] : r_begin + a.n_nodes) - r_begin;
  if (a.plan && a.plan_cap == 0 && tid == 0) {
    const int prev = g > a.g_base ? (a.graph_off ? a.graph_off[g - 1] : (g - 1) * a.n_nodes) : -1;
    for (int w = prev < 0 ? 0 : prev / a.plan_capp + 1; w <= r_begin / a.plan_capp && w <= a.plan_n; ++w) a.plan[w] = g;
    if (blockIdx.x == (unsigned)a.n_graphs - 1)
      for (int w = r_begin / a.plan_capp + 1; w <= a.plan_n; ++w) a.plan[w] = a.g_base + a.n_graphs;
  }
  if (n > a.rows_cap || n < 0) { if (tid == 0 && a.err) atomicOr(a.err, 1); return; }
  int* sR = reinterpret_cast<int*>(sD + a.rows_cap * a.mask_words);         // [rows_cap + 1] the graph's slice of row_ptr
  for (int i = tid; i <= n; i += 256) sR[i] = a.row_ptr[r_begin + i];
  for (int i = tid; i < n * a.mask_words; i += 256) sM[i] = 0u;
  __syncthreads();
  // 32 threads per destination row, AD_MH passes of 8 rows in flight at once with all 4 col_idx loads of each: a 128-link
  // graph is two round trips to memory (round 3: two passes in flight, eight round trips -- the launch lasts as long as
  // its largest graph's chain).  The row bounds come from the LDS copy of row_ptr.
  constexpr int AD_MH = 8;
  for (int i0 = tid; i0 < n * 32; i0 += 256 * AD_MH) {
    int p[AD_MH][4], q[AD_MH];
#pragma unroll
    for (int h = 0; h < AD_MH; ++h) {
      const int i = i0 + 256 * h;
      q[h] = min(i >> 5, n - 1);
      const int e0 = sR[q[h]] + (i & 31), e1 = i < n * 32 ? sR[q[h] + 1] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) p[h][u] = e0 + 32 * u < e1 ? a.col_idx[e0 + 32 * u] : -1;
    }
#pragma unroll
    for (int h = 0; h < AD_MH; ++h) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p[h][u] >= 0 && p[h][u] < n) atomicOr(&sM[p[h][u] * a.mask_words + (q[h] >> 5)], 1u << (q[h] & 31));
      const int i = i0 + 256 * h;
      if (i < n * 32)
        for (int e = sR[q[h]] + (i & 31) + 128; e < sR[q[h] + 1]; e += 32) {
          const int pp = a.col_idx[e];
          if (pp >= 0 && pp < n) atomicOr(&sM[pp * a.mask_words + (q[h] >> 5)], 1u << (q[h] & 31));
        }
    }
  }
  __syncthreads();
  // by destination = the transpose of the bit matrix.  A wave takes a 32-destination word column: lane p reads ITS row's
  // word, the two 32 x 32 bit blocks of the wave (sources 0-31 / 32-63 of the block) are transposed in place by five
  // exchange steps across lanes (blocks of 16, 8, 4, 2, 1: lane ^ j hands over the half this lane lacks), and lane t ends up
  // holding 32 source bits of destination 32 qw + t -- one word of its by-destination row.  (Round 3: one LDS read and one
  // ballot per destination and source block; this launch is bound by its integer instruction count, not by memory.)
  // Words past the graph's own sources stay unwritten: every reader masks a row to its graph's n bits.
  {
    const int lane = tid & 63, wv = tid >> 6, n_blk = (n + 63) >> 6;
    for (int qw = wv; 32 * qw < n; qw += 4)
      for (int blk = 0; blk < n_blk; ++blk) {
        const int pp = 64 * blk + lane;
        unsigned w = pp < n ? sM[pp * a.mask_words + qw] : 0u;
        unsigned mk = 0x0000ffffu;
#pragma unroll
        for (int j = 16; j; j >>= 1) {
          const unsigned other = (unsigned)__shfl_xor((int)w, j);
          w = (lane & j) ? ((w & (mk << j)) | ((other >> j) & mk)) : ((w & mk) | ((other & mk) << j));
          mk ^= mk << (j >> 1);
        }
        const int q = 32 * qw + (lane & 31), word = 2 * blk + (lane >> 5);
        if (q < n && word < a.mask_words) sD[q * a.mask_words + word] = w;
      }
  }
  __syncthreads();
  for (int i = tid; i < n * a.mask_words; i += 256) {
    a.adj[(int64_t)r_begin * a.mask_words + i] = sM[i];
    a.adjT[(int64_t)r_begin * a.mask_words + i] = sD[i];
  }
}

// LDS layout of the [rows][64] feature tile: NO padding, row r is rotated by 16 (r & 3) floats instead -- the four
// k-groups of a wave (rows 4 s + kg of a k-step) then hit disjoint quarters of the 64 banks exactly as the 80-float row
// stride of round 2 made them, float4 accesses stay aligned, and a graph of 100 / 128 links takes 34 / 39 KB instead of
// 41 / 46: FOUR workgroups per CU instead of three.  These launches are latency chains per workgroup (profiles/HISTORY.md 3b): the
// fourth resident workgroup is worth 73.2 -> 59.3 / 96.8 -> 80.0 us at configs[3] and 30.7 -> 26.7 / 41.4 -> 36.8 us at configs[4].
constexpr int AD_LDT = 64;
__device__ __forceinline__ int ad_off(int row, int col) { return row * AD_LDT + ((col + ((row & 3) << 4)) & 63); }

// grid = n_graphs * n_fg workgroups.  Workgroup ids are dealt round-robin to the 8 XCDs, so the n_fg feature
// groups of one graph are given ids with the SAME id mod 8: they run on one XCD and share its L2.
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void k_agg_dense(AggDenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sT = smem;                                                         // [rows_cap][AD_LDT]
  unsigned* sM = reinterpret_cast<unsigned*>(sT + a.rows_cap * AD_LDT);     // [rows_cap][mask_words]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  const int b = blockIdx.x, span = 8 * a.n_fg;
  const int gl = (b / span) * 8 + (b & 7), fg = (b % span) >> 3;
  if (gl >= a.n_graphs) return;
  const int g = a.g_base + gl, f0 = fg << 6;
  const int r_begin = a.graph_off ? a.graph_off[g] : g * a.n_nodes;
  const int n = (a.graph_off ? a.graph_off[g + 1] : r_begin + a.n_nodes) - r_begin;
  if (n > a.rows_cap || n < 1) { if (tid == 0 && a.err) atomicOr(a.err, 1); return; }

  // the graph's non-edges decide the form (uniform over the workgroup; the feature groups of a graph agree)
  const int n_edges = a.row_ptr[r_begin + n] - a.row_ptr[r_begin];
  const bool compl_form = n * n - n_edges <= AD_COMPL_PER_ROW * n;
  // stage the graph's [n][64] feature slice (4 loads in flight per thread) and its adjacency bit masks: by source for the
  // MFMA product and for the transposed complement walk, by destination for the forward complement walk
  const int n_mw = n * a.mask_words;
  const unsigned* msrc = (compl_form && !TRANSPOSE) ? a.adjT : a.adj;
  for (int i = tid; i < a.rows_cap * a.mask_words; i += 256) sM[i] = i < n_mw ? msrc[(int64_t)r_begin * a.mask_words + i] : 0u;
  for (int i0 = 0; i0 < n * 16; i0 += 2048) {                              // 8 loads in flight per thread: 128 rows per pass
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u * 256 + tid, n * 16 - 1);
      v[u] = *reinterpret_cast<const float4*>(a.src + (int64_t)(r_begin + (i >> 4)) * a.src_stride + f0 + ((i & 15) << 2));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid;
      if (i < n * 16) *reinterpret_cast<float4*>(sT + ad_off(i >> 4, (i & 15) << 2)) = v[u];
    }
  }
  __syncthreads();

  if (compl_form) {
    // out[r] = S - sum over the zero bits p < n of row r's mask of tile[p]   (forward: r = destination, mask by
    // destination; transpose: r = source, mask by source -- the same walk)
    float* sP = reinterpret_cast<float*>(sM + a.rows_cap * a.mask_words);   // [4 waves][64] partial column sums
    const int c = tid & 15, rg = tid >> 4;                                  // float4 column, row group (4 per wave)
    f32x4 S;
    {
      f32x4 ps = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int r = rg; r < n; r += 16) ps += *reinterpret_cast<const f32x4*>(sT + ad_off(r, 4 * c));
      // the wave's four row groups through lane exchanges, the four waves through 1 KB of LDS (16 partial rows there would
      // cost the fifth resident workgroup at 100 links): ((g0 + g1) + (g2 + g3)) per wave, ((w0 + w1) + (w2 + w3)) -- fixed
#pragma unroll
      for (int e = 0; e < 4; ++e) ps[e] += __shfl_xor(ps[e], 16, 64);
#pragma unroll
      for (int e = 0; e < 4; ++e) ps[e] += __shfl_xor(ps[e], 32, 64);
      if ((lane >> 4) == 0) *reinterpret_cast<f32x4*>(sP + wv * 64 + 4 * c) = ps;
      __syncthreads();
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(sP + 0 * 64 + 4 * c), w1 = *reinterpret_cast<const f32x4*>(sP + 1 * 64 + 4 * c);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(sP + 2 * 64 + 4 * c), w3 = *reinterpret_cast<const f32x4*>(sP + 3 * 64 + 4 * c);
      S = (w0 + w1) + (w2 + w3);
    }
    for (int r0 = 0; r0 < n; r0 += 16) {
      const int r = r0 + rg;
      if (r >= n) break;
      const int64_t grow = r_begin + r;
      const f32x4 addv = a.add ? ld4(a.add + grow * a.add_stride + f0 + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 gate = a.mask ? ld4(a.mask + grow * a.F + f0 + 4 * c) : (f32x4){1.f, 1.f, 1.f, 1.f};
      // Per row: through the complement only when the row HAS fewer non-edges than edges; a row with few (or no) in-neighbours
      // inside an otherwise dense graph -- every graph of <= 8 links takes this form -- sums its edges directly, so that
      // its result carries the rounding of its own terms, not of the whole column sum (an isolated node gets an exact 0).
      int ones = 0;
      for (int w = 0; w < a.mask_words; ++w) {
        const int left = n - 32 * w;
        if (left <= 0) break;
        ones += __builtin_popcount(sM[r * a.mask_words + w] & (left >= 32 ? 0xffffffffu : ((1u << left) - 1u)));
      }
      const bool direct = 2 * ones < n;
      f32x4 miss = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < a.mask_words; ++w) {
        const int left = n - 32 * w;                                        // valid bits of this word
        if (left <= 0) break;
        const unsigned m = sM[r * a.mask_words + w];
        unsigned z = (direct ? m : ~m) & (left >= 32 ? 0xffffffffu : ((1u << left) - 1u));
        while (z) {
          const int p = 32 * w + __builtin_ctz(z);
          z &= z - 1;
          miss += *reinterpret_cast<const f32x4*>(sT + ad_off(p, 4 * c));
        }
      }
      f32x4 v = (direct ? miss : S - miss) + addv;
      v = gate4(v, gate);
      st4(a.out + grow * a.F + f0 + 4 * c, v);
    }
    return;
  }

  const int n_rt = (n + 15) >> 4;
  const int n_k4 = (n + 15) >> 4;                                           // groups of 4 k-steps (16 contraction rows)
  for (int rt = wv; rt < n_rt; rt += 4) {
    const int row = rt * 16 + j;                                            // this lane's OUTPUT row (dest q / source p)
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // epilogue operands first (HBM latency hidden behind the MFMA loop)
    const bool valid = row < n;
    const int64_t grow = r_begin + (valid ? row : 0);
    f32x4 addv[4], gate[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      addv[nt] = a.add ? ld4(a.add + grow * a.add_stride + f0 + nt * 16 + 4 * kg) : (f32x4){0.f, 0.f, 0.f, 0.f};
      gate[nt] = a.mask ? ld4(a.mask + grow * a.F + f0 + nt * 16 + 4 * kg) : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll 1
    for (int k4 = 0; k4 < n_k4; ++k4) {
      float bv[4], av[4][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {                                         // all LDS reads of 4 k-steps, then 16 MFMAs
        const int kk = k4 * 16 + s * 4 + kg;                                // contraction index of this lane (< rows_cap)
        const int kc = min(kk, n - 1);                                      // clamped: finite value x 0.0
        unsigned w;
        if (!TRANSPOSE) w = sM[kk * a.mask_words + (row >> 5)] >> (row & 31);   // Adj[p = kk][q = row]
        else w = sM[row * a.mask_words + (kk >> 5)] >> (kk & 31);               // Adj[p = row][q = kk]
        bv[s] = (float)(w & 1u);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) av[s][nt] = sT[ad_off(kc, nt * 16 + j)];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = V2X_MFMA(av[s][nt], bv[s], acc[nt]);
    }
    if (valid) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x4 v = acc[nt] + addv[nt];
        v = gate4(v, gate[nt]);
        st4(a.out + grow * a.F + f0 + nt * 16 + 4 * kg, v);
      }
    }
  }
}

}  // namespace v2x
