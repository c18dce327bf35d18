// The decision MLP of a fit step for the SHARES of the metric's global batch, one wave per (node slot, 16-row tile), weights
// STREAMED from a padded image in global memory (round 6).
//
// k_mlp_train_wg (kernels_mlpwg.hpp) stages a slot's four weight images in LDS once per workgroup and keeps the Dense layers'
// weight gradients in 272 accumulators: right when a wave walks 5-6 tiles (batch 4096).  At the 512- / 1024-graph shares a wave
// has ONE or TWO tiles: the launch is staging (3.1 us) + a tile (11.9) + the accumulator exchange and slab write (3) on 160 of 256
// CUs, and every weight a wave reads from LDS it reads exactly once per tile.  Here the wave reads them straight from L2 in the
// order it needs them, a few k-blocks ahead of the MFMAs (no LDS, no barrier, 64-thread workgroups that spread over all CUs),
// writes what k_mlp_train writes for the weight-gradient launch (z1..z3, dq, dz3, dz2, the gated dz1 rows) and leaves ALL four
// Dense weight gradients to roles of the graph layers' launch (k_wgrad MODE 4: the Dense-0 halves of round 6 + Dense 1..3).
//
// Arithmetic: the chain of k_mlp_train (TF autodiff of the K.dot's of BS_brain.py:176-179, Huber :86-87), same k-block order per
// accumulator -- q, the loss rows and the data gradients are bitwise those of k_mlp_train / k_mlp_train_wg; the weight gradients
// differ from k_mlp_train_wg's by the order of the sum over rows only (as the Dense-0 roles do).
//
// The image: MlpFrag<F> below, one per slot (135 KB), written by k_mlp_image from the flat parameters.
#pragma once
#include "kernels.hpp"
#include "kernels_mlpwg.hpp"

namespace v2x {

// Fragment-major image of one slot's Dense layers: per (layer, direction, k-block, n-tile) one 1 KiB block = the float4 every lane
// of the wave feeds to the four MFMA k-steps of that tile (forward: W[16 kb + 4 kg + s][16 nt + j], s = 0..3; reverse:
// W[orow(nt) + j][16 kb + 4 kg ..+3]) -- one fully coalesced global_load_dwordx4 per tile.  (The first version read MlpLds' row-major
// image: 272 dword loads per wave for the forward weights alone against the 64 vector-memory instructions a wave may have in
// flight -- 21.8 us per launch at the 512-graph share, no faster than k_mlp_train_wg.)
template <int F>
struct MlpFrag {
  static constexpr int FB = F / 16, KB1 = 2 * FB + 1;
  static constexpr int FW1 = 0, FW2 = FW1 + KB1 * 5, FW3 = FW2 + 5 * 3, FW4 = FW3 + 3 * 2;      // (in 256-float blocks)
  static constexpr int BW4 = FW4 + 2 * 1, BW3 = BW4 + 1 * 2, BW2 = BW3 + 2 * 3, BW1 = BW2 + 3 * 5;
  static constexpr int BLOCKS = BW1 + 5 * 2 * FB;
  static constexpr int B1 = BLOCKS * 256, B2 = B1 + H1, B3 = B2 + H2P, B4 = B3 + H3P;
  static constexpr int TOTAL = B4 + CP;                                                          // floats per slot (a multiple of 4)
};

template <int F>
__global__ __launch_bounds__(256) void k_mlp_image(MlpArgs a, float* img) {
  using L = MlpLds<F>;
  using G = MlpFrag<F>;
  constexpr int FB = F / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int slot = blockIdx.x;
  mlp_fill_lds<F>(smem, a, slot, true, threadIdx.x);
  __syncthreads();
  float* dst = img + (int64_t)slot * G::TOTAL;
  for (int i = threadIdx.x; i < G::BLOCKS * 64; i += 256) {
    const int blk = i >> 6, lane = i & 63, j = lane & 15, kg = lane >> 4;
    float4 v;
    if (blk < G::BW4) {                                   // forward fragments
      const float* sW; int ld, nt_n, b;
      if (blk < G::FW2) { sW = smem + L::W1; ld = LD1; nt_n = 5; b = blk - G::FW1; }
      else if (blk < G::FW3) { sW = smem + L::W2; ld = LD2; nt_n = 3; b = blk - G::FW2; }
      else if (blk < G::FW4) { sW = smem + L::W3; ld = LD3; nt_n = 2; b = blk - G::FW3; }
      else { sW = smem + L::W4; ld = LD4; nt_n = 1; b = blk - G::FW4; }
      const int kb = b / nt_n, nt = b - kb * nt_n;
      const float* p = sW + (kb * 16 + 4 * kg) * ld + nt * 16 + j;
      v = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
    } else {                                              // reverse fragments
      const float* sW; int ld, nt_n, b; bool skip = false;
      if (blk < G::BW3) { sW = smem + L::W4; ld = LD4; nt_n = 2; b = blk - G::BW4; }
      else if (blk < G::BW2) { sW = smem + L::W3; ld = LD3; nt_n = 3; b = blk - G::BW3; }
      else if (blk < G::BW1) { sW = smem + L::W2; ld = LD2; nt_n = 5; b = blk - G::BW2; }
      else { sW = smem + L::W1; ld = LD1; nt_n = 2 * FB; b = blk - G::BW1; skip = true; }
      const int kb = b / nt_n, nt = b - kb * nt_n;
      const int orow = skip ? (nt < FB ? nt * 16 : F + XE + (nt - FB) * 16) : nt * 16;
      v = *reinterpret_cast<const float4*>(sW + (orow + j) * ld + kb * 16 + 4 * kg);
    }
    *reinterpret_cast<float4*>(dst + (int64_t)i * 4) = v;
  }
  for (int i = threadIdx.x; i < H1 + H2P + H3P + CP; i += 256) dst[G::B1 + i] = smem[L::B1 + i];
}

template <int F>
constexpr int mlp_image_floats() { return MlpFrag<F>::TOTAL; }

template <int NT>
__device__ __forceinline__ void fwd_frag(const float* base, int kb, int lane, float (&w)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float4 t = *reinterpret_cast<const float4*>(base + ((kb * NT + nt) * 64 + lane) * 4);
    w[nt][0] = t.x; w[nt][1] = t.y; w[nt][2] = t.z; w[nt][3] = t.w;
  }
}
template <int NT>
__device__ __forceinline__ void bwd_frag(const float* base, int kb, int lane, float4 (&t)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) t[nt] = *reinterpret_cast<const float4*>(base + ((kb * NT + nt) * 64 + lane) * 4);
}

// one 16-row tile of one slot in one wave.  z0 = the lane's float4 of [h | x | agg] per k-block (lane (j, kg): row j, columns
// 16 kb + 4 kg ..), yv = the row's targets; go / gs: where the lane's [dh | dagg] float4 go (fragment- or row-major).
template <int F>
__device__ __forceinline__ void mlp_stream_tile(const MlpArgs& a, const float* img, const f32x4 (&z0)[2 * (F / 16) + 1], const f32x4 yv,
                                                const int64_t row, const int64_t srow, const bool valid, float* gha_lane, const int gs,
                                                const int j, const int kg) {
  using L = MlpFrag<F>;
  constexpr int FB = F / 16, KB1 = 2 * FB + 1;
  const int lane = j + 16 * kg;
  const float* F1 = img + L::FW1 * 256;
  const float* F2 = img + L::FW2 * 256;
  const float* F3 = img + L::FW3 * 256;
  const float* F4 = img + L::FW4 * 256;
  const float* R4 = img + L::BW4 * 256;
  const float* R3 = img + L::BW3 * 256;
  const float* R2 = img + L::BW2 * 256;
  const float* R1 = img + L::BW1 * 256;
  // ---- requests, oldest first = needed first (loads return in order): Dense-0's first PF k-blocks, then all of Dense-1
  constexpr int PF = 6;
  float w5[PF][5][4];
#pragma unroll
  for (int p = 0; p < PF; ++p) fwd_frag<5>(F1, p, lane, w5[p]);
  float w3[5][3][4];
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) fwd_frag<3>(F2, kb, lane, w3[kb]);
  f32x4 bias1[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) bias1[nt] = ld4(img + L::B1 + nt * 16 + 4 * kg);
  // ================= forward
  f32x4 z1[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) z1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB1; ++kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) z1[nt] = V2X_MFMA(w5[kb % PF][nt][s], z0[kb][s], z1[nt]);
    if (kb + PF < KB1) fwd_frag<5>(F1, kb + PF, lane, w5[kb % PF]);
  }
  // Dense 2, 3 forward and Dense 3, 2 reverse: requested while Dense-1 runs
  float w2[3][2][4], w1[2][1][4];
  float4 t2[1][2], t3[2][3];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) fwd_frag<2>(F3, kb, lane, w2[kb]);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) fwd_frag<1>(F4, kb, lane, w1[kb]);
  f32x4 bias2[3], bias3[2], bias4;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) bias2[nt] = ld4(img + L::B2 + nt * 16 + 4 * kg);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bias3[nt] = ld4(img + L::B3 + nt * 16 + 4 * kg);
  bias4 = ld4(img + L::B4 + 4 * kg);
  bwd_frag<2>(R4, 0, lane, t2[0]);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) bwd_frag<3>(R3, kb, lane, t3[kb]);
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    z1[nt] = relu4(z1[nt] + bias1[nt]);
    if (valid) st4(a.z1 + row * H1 + nt * 16 + 4 * kg, z1[nt]);
  }
  f32x4 z2[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) z2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 5; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) z2[nt] = V2X_MFMA(w3[kb][nt][s], z1[kb][s], z2[nt]);
  // Dense-1 reverse: requested while Dense 2, 3 run
  float4 t5[3][5];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) bwd_frag<5>(R2, kb, lane, t5[kb]);
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    z2[nt] = relu4(z2[nt] + bias2[nt]);
    if (valid && nt * 16 + 4 * kg < H2) st4(a.z2 + srow * H2 + nt * 16 + 4 * kg, z2[nt]);
  }
  f32x4 z3[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) z3[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) z3[nt] = V2X_MFMA(w2[kb][nt][s], z2[kb][s], z3[nt]);
  // Dense-0 reverse, first PF8 k-blocks: requested before the short chains in the middle
  constexpr int PF8 = 5;
  float4 t8[PF8][2 * FB];
#pragma unroll
  for (int p = 0; p < PF8; ++p) bwd_frag<2 * FB>(R1, p, lane, t8[p]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    z3[nt] = relu4(z3[nt] + bias3[nt]);
    if (valid && nt * 16 + 4 * kg < H3) st4(a.z3 + srow * H3 + nt * 16 + 4 * kg, z3[nt]);
  }
  f32x4 qa = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s) qa = V2X_MFMA(w1[kb][0][s], z3[kb][s], qa);
  const f32x4 qv = qa + bias4;
  if (valid && 4 * kg < a.C) st4(a.q + row * a.C + 4 * kg, qv);
  // ================= Huber (delta = 1); rows past the end carry a zero gradient through everything below
  f32x4 g4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (4 * kg < a.C) {
    float ls = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float err = qv[c] - yv[c];
      const float ab = fabsf(err), quad = fminf(ab, 1.f);
      ls += 0.5f * quad * quad + (ab - quad);
      g4[c] = valid ? fminf(fmaxf(err, -1.f), 1.f) * a.inv_denom : 0.f;
    }
    if (valid) {
      st4(a.dq + srow * a.C + 4 * kg, g4);
      a.rowloss[srow] = ls;
    }
  }
  // ================= reverse
  f32x4 d3[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) d3[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) d3[nt] = V2X_MFMA(t2[0][nt].x, g4[0], d3[nt]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) d3[nt] = V2X_MFMA(t2[0][nt].y, g4[1], d3[nt]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) d3[nt] = V2X_MFMA(t2[0][nt].z, g4[2], d3[nt]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) d3[nt] = V2X_MFMA(t2[0][nt].w, g4[3], d3[nt]);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const bool in_range = nt * 16 + 4 * kg < H3;
    d3[nt] = in_range ? gate4(d3[nt], z3[nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (valid && in_range) st4(a.dz3 + srow * H3 + nt * 16 + 4 * kg, d3[nt]);
  }
  f32x4 d2[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) d2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt] = V2X_MFMA(t3[kb][nt].x, d3[kb][0], d2[nt]);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt] = V2X_MFMA(t3[kb][nt].y, d3[kb][1], d2[nt]);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt] = V2X_MFMA(t3[kb][nt].z, d3[kb][2], d2[nt]);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt] = V2X_MFMA(t3[kb][nt].w, d3[kb][3], d2[nt]);
  }
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const bool in_range = nt * 16 + 4 * kg < H2;
    d2[nt] = in_range ? gate4(d2[nt], z2[nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (valid && in_range) st4(a.dz2 + srow * H2 + nt * 16 + 4 * kg, d2[nt]);
  }
  f32x4 d1[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) d1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt] = V2X_MFMA(t5[kb][nt].x, d2[kb][0], d1[nt]);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt] = V2X_MFMA(t5[kb][nt].y, d2[kb][1], d1[nt]);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt] = V2X_MFMA(t5[kb][nt].z, d2[kb][2], d1[nt]);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt] = V2X_MFMA(t5[kb][nt].w, d2[kb][3], d1[nt]);
  }
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    d1[nt] = gate4(d1[nt], z1[nt]);
    if (valid) st4(a.dz1 + row * H1 + nt * 16 + 4 * kg, d1[nt]);
  }
  f32x4 o[2 * FB];
#pragma unroll
  for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) {
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = V2X_MFMA(t8[kb % PF8][nt].x, d1[kb][0], o[nt]);
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = V2X_MFMA(t8[kb % PF8][nt].y, d1[kb][1], o[nt]);
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = V2X_MFMA(t8[kb % PF8][nt].z, d1[kb][2], o[nt]);
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = V2X_MFMA(t8[kb % PF8][nt].w, d1[kb][3], o[nt]);
    if (kb + PF8 < 5) bwd_frag<2 * FB>(R1, kb + PF8, lane, t8[kb % PF8]);
  }
#pragma unroll
  for (int nt = 0; nt < 2 * FB; ++nt)
    if (valid) st4(gha_lane + nt * gs, o[nt]);
}

// grid: one 64-thread workgroup per (slot, tile); unit u -> slot = u / tiles_per_slot.  Workgroups go to the 8 XCDs round-robin
// by id: with units dealt so that an XCD owns a contiguous run of them, a slot's image is read into ONE or two L2s instead of 8.
template <int F, bool FRAG>
__global__ __launch_bounds__(64) void k_mlp_stream(MlpArgs a, const float* img, int tiles_per_slot, int n_units) {
  constexpr int FB = F / 16, KB1 = 2 * FB + 1;
  const int lane = threadIdx.x, j = lane & 15, kg = lane >> 4;
  int u = blockIdx.x;
  if ((n_units & 7) == 0) u = (u & 7) * (n_units >> 3) + (u >> 3);
  const int slot = u / tiles_per_slot, t = u - slot * tiles_per_slot;
  const int idx = min(t * 16 + j, a.n_idx - 1);
  const bool valid = t * 16 + j < a.n_idx;
  const int64_t row = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
  const int64_t srow = (int64_t)slot * a.srow_stride + a.idx_base + idx;
  const int cq = 4 * kg < a.C ? 4 * kg : 0;
  f32x4 z0[KB1];
  const int64_t ho = FRAG ? ((int64_t)slot * a.frag_groups + min(t, a.frag_groups - 1)) * (FB * 256) + lane * 4 : row * F + 4 * kg;
  constexpr int hs = FRAG ? 256 : 16;
#pragma unroll
  for (int b = 0; b < FB; ++b) z0[b] = ld4(a.h + ho + b * hs);
  z0[FB] = ld4(a.xe + row * XE + 4 * kg);
#pragma unroll
  for (int b = 0; b < FB; ++b) z0[FB + 1 + b] = ld4(a.agg + ho + b * hs);
  const f32x4 yv = ld4(a.y + row * a.C + cq);
  const int64_t go = FRAG ? ((int64_t)slot * a.frag_groups + min(t, a.frag_groups - 1)) * (2 * FB * 256) + lane * 4 : row * (2 * F) + 4 * kg;
  mlp_stream_tile<F>(a, img + (int64_t)slot * mlp_image_floats<F>(), z0, yv, row, srow, valid, a.gha + go, FRAG ? 256 : 16, j, kg);
}

}  // namespace v2x
