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

constexpr int WD_TM = 128, WD_TN = 64, WD_KC = 16, WD_LDA = WD_TN + 4, WD_LDB = WD_KC + 4;

template <bool TRANS>
__global__ __launch_bounds__(256) void k_wide_gemm(WideGemmArgs a) {
  __shared__ __attribute__((aligned(16))) float sA[2][WD_KC * WD_LDA];   // weights  [k][n]
  __shared__ __attribute__((aligned(16))) float sB[2][WD_TM * WD_LDB];   // activations [row][k]
  const int slot = blockIdx.z, m0 = blockIdx.x * WD_TM, n0 = blockIdx.y * WD_TN;
  const float* Wg = a.W + slot * a.slot_stride;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;

  // global -> register staging roles
  const int rB = tid >> 2, cB = (tid & 3) << 2;                 // activation rows rB and rB + 64, 4 k-columns
  const int64_t rowg0 = (int64_t)(a.idx_base + min(m0 + rB, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
  const int64_t rowg1 = (int64_t)(a.idx_base + min(m0 + rB + 64, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
  // weights, forward: k row tid>>4, 4 output columns; transposed: output column tid>>2, 4 k values
  const int kaF = tid >> 4, naF = (tid & 15) << 2;
  const int naT = tid >> 2, kaT = (tid & 3) << 2;
  const int colF = n0 + naF;
  const bool okF = colF < a.n_real;
  const int nT = n0 + naT;
  const bool okT = nT < a.n_out;
  const int wrowT = okT ? (nT < a.split ? nT : nT + a.skip) : 0;

  const int n_chunks = a.k_total / WD_KC;
  auto gload = [&](int kc, float4& va, float4& vb0, float4& vb1) {
    int kcol = kc * WD_KC;
    const float* p = a.seg[0].ptr; int st = a.seg[0].stride;
    if (a.n_seg > 1 && kcol >= a.seg[0].width) {
      kcol -= a.seg[0].width; p = a.seg[1].ptr; st = a.seg[1].stride;
      if (a.n_seg > 2 && kcol >= a.seg[1].width) { kcol -= a.seg[1].width; p = a.seg[2].ptr; st = a.seg[2].stride; }
    }
    vb0 = *reinterpret_cast<const float4*>(p + rowg0 * st + kcol + cB);
    vb1 = *reinterpret_cast<const float4*>(p + rowg1 * st + kcol + cB);
    if (!TRANS) {
      const int rr = real_row(a.pad, kc * WD_KC + kaF);
      const bool ok = okF && rr >= 0;
      const float4 t = *reinterpret_cast<const float4*>(Wg + (ok ? (int64_t)rr * a.n_real + colF : 0));
      const float mk = ok ? 1.f : 0.f;
      va = make_float4(t.x * mk, t.y * mk, t.z * mk, t.w * mk);
    } else {
      const float4 t = *reinterpret_cast<const float4*>(Wg + (int64_t)wrowT * a.n_real + kc * WD_KC + kaT);
      const float mk = okT ? 1.f : 0.f;
      va = make_float4(t.x * mk, t.y * mk, t.z * mk, t.w * mk);
    }
  };
  auto lstore = [&](int buf, const float4& va, const float4& vb0, const float4& vb1) {
    *reinterpret_cast<float4*>(&sB[buf][rB * WD_LDB + cB]) = vb0;
    *reinterpret_cast<float4*>(&sB[buf][(rB + 64) * WD_LDB + cB]) = vb1;
    if (!TRANS) {
      *reinterpret_cast<float4*>(&sA[buf][kaF * WD_LDA + naF]) = va;
    } else {
      sA[buf][(kaT + 0) * WD_LDA + naT] = va.x; sA[buf][(kaT + 1) * WD_LDA + naT] = va.y;
      sA[buf][(kaT + 2) * WD_LDA + naT] = va.z; sA[buf][(kaT + 3) * WD_LDA + naT] = va.w;
    }
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 va, vb0, vb1;
  gload(0, va, vb0, vb1);
  lstore(0, va, vb0, vb1);
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc < n_chunks; ++kc) {
    const int buf = kc & 1;
    const bool more = kc + 1 < n_chunks;
    if (more) gload(kc + 1, va, vb0, vb1);
    float w[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int s = 0; s < 4; ++s) w[nt][s] = sA[buf][(4 * kg + s) * WD_LDA + nt * 16 + j];
    f32x4 b[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) b[rt] = ld4(&sB[buf][(32 * wv + 16 * rt + j) * WD_LDB + 4 * kg]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[rt][nt] = V2X_MFMA(w[nt][s], b[rt][s], acc[rt][nt]);
    if (more) lstore(buf ^ 1, va, vb0, vb1);
    __syncthreads();
  }

  // epilogue: lane holds out[row 32*wv + 16*rt + j][n0 + nt*16 + 4*kg .. +3]
  const float* bias = Wg + (int64_t)a.pad.k_real * a.n_real;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int idx = m0 + 32 * wv + 16 * rt + j;
    if (idx >= a.n_idx) continue;
    const int64_t rowg = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = n0 + nt * 16 + 4 * kg;
      if (col >= a.n_out) continue;
      f32x4 v = acc[rt][nt];
      if (!TRANS && a.has_bias) v = v + ld4(bias + col);
      if (!TRANS && a.relu) v = relu4(v);
      st4(a.out + rowg * a.out_stride + col, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
struct WideWgradArgs {
  WideSeg seg[3]; int n_seg;             // K operand; a 64-wide K tile never straddles two segments
  int seg_kpad[3];                       // padded K row where each segment starts
  const float* dpre; int d_stride; int n_real;
  RowPad pad;
  float* slab; int64_t slab_stride;      // slab[split][P]
  int64_t layer_off, slot_stride;
  int n_idx, row_stride, base_mul, idx_base;
  int n_split, rows_per_split;           // rows_per_split is a multiple of 32
};

constexpr int WW_TR = 32, WW_LD = 64 + 4;

__global__ __launch_bounds__(256) void k_wide_wgrad(WideWgradArgs a) {
  __shared__ __attribute__((aligned(16))) float sX[2][WW_TR * WW_LD];
  __shared__ __attribute__((aligned(16))) float sD[2][WW_TR * WW_LD];
  const int slot = blockIdx.z / a.n_split, sp = blockIdx.z - slot * a.n_split;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, kg = lane >> 4;
  // K tile -> (segment, column offset, valid width)
  int kt = blockIdx.x;
  const float* xp = a.seg[0].ptr; int xst = a.seg[0].stride, xw = a.seg[0].width, kpad0 = a.seg_kpad[0];
  {
    const int t0 = (a.seg[0].width + 63) >> 6;
    if (a.n_seg > 1 && kt >= t0) {
      kt -= t0; xp = a.seg[1].ptr; xst = a.seg[1].stride; xw = a.seg[1].width; kpad0 = a.seg_kpad[1];
      const int t1 = (a.seg[1].width + 63) >> 6;
      if (a.n_seg > 2 && kt >= t1) { kt -= t1; xp = a.seg[2].ptr; xst = a.seg[2].stride; xw = a.seg[2].width; kpad0 = a.seg_kpad[2]; }
    }
  }
  const int kcol0 = kt * 64, kw = min(64, xw - kcol0);
  kpad0 += kcol0;
  const int n0 = blockIdx.y * 64;
  const int i_begin = sp * a.rows_per_split, i_end = min(i_begin + a.rows_per_split, a.n_idx);
  const int n_chunks = (max(i_end - i_begin, 0) + WW_TR - 1) / WW_TR;

  const int rL = tid >> 4, cL = (tid & 15) << 2;                // rows rL and rL + 16, 4 columns
  const bool okx = cL < kw, okd = n0 + cL < a.n_real;
  const int xcol = kcol0 + (okx ? cL : 0), dcol = okd ? n0 + cL : 0;
  auto gload = [&](int c, float4 (&vx)[2], float4 (&vd)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = i_begin + c * WW_TR + rL + 16 * h;
      const float mrow = idx < i_end ? 1.f : 0.f;
      const int64_t rowg = (int64_t)(a.idx_base + min(idx, a.n_idx - 1)) * a.row_stride + slot * a.base_mul;
      const float4 x = *reinterpret_cast<const float4*>(xp + rowg * xst + xcol);
      const float4 d = *reinterpret_cast<const float4*>(a.dpre + rowg * a.d_stride + dcol);
      const float mx = okx ? 1.f : 0.f, md = okd ? mrow : 0.f;
      vx[h] = make_float4(x.x * mx, x.y * mx, x.z * mx, x.w * mx);
      vd[h] = make_float4(d.x * md, d.y * md, d.z * md, d.w * md);
    }
  };
  auto lstore = [&](int buf, const float4 (&vx)[2], const float4 (&vd)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(&sX[buf][(rL + 16 * h) * WW_LD + cL]) = vx[h];
      *reinterpret_cast<float4*>(&sD[buf][(rL + 16 * h) * WW_LD + cL]) = vd[h];
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;                                             // column n0 + tid (threads < 64 of the kt == 0 tile)
  const bool do_bias = blockIdx.x == 0 && tid < 64;

  float4 vx[2], vd[2];
  if (n_chunks > 0) { gload(0, vx, vd); lstore(0, vx, vd); }
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < n_chunks;
    if (more) gload(c + 1, vx, vd);
#pragma unroll
    for (int s = 0; s < WW_TR / 4; ++s) {
      const float xa = sX[buf][(4 * s + kg) * WW_LD + 16 * wv + j];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = V2X_MFMA(xa, sD[buf][(4 * s + kg) * WW_LD + nt * 16 + j], acc[nt]);
    }
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < WW_TR; ++r) bsum += sD[buf][r * WW_LD + tid];
    }
    if (more) lstore(buf ^ 1, vx, vd);
    __syncthreads();
  }

  // lane holds dW[kpad0 + 16*wv + 4*kg + r][n0 + nt*16 + j]
  float* dst = a.slab + (int64_t)sp * a.slab_stride + a.layer_off + slot * a.slot_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = 16 * wv + 4 * kg + r;
    const int rr = kk < kw ? real_row(a.pad, kpad0 + kk) : -1;
    if (rr < 0) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = n0 + nt * 16 + j;
      if (col < a.n_real) dst[(int64_t)rr * a.n_real + col] = acc[nt][r];
    }
  }
  if (do_bias && n0 + tid < a.n_real) dst[(int64_t)a.pad.k_real * a.n_real + n0 + tid] = bsum;
}

}  // namespace v2x
