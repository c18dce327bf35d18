// k_mlp_train_wg: the decision MLP of a fit step with its four weight gradients in the SAME pass over the rows.
//
// Replaces, for narrow features, k_mlp_train + the four Dense roles of k_wgrad (TF autodiff of the K.dot's of
// BS_brain.py:176-179 under keras' fit, :231-239).  In that split the hidden activations z1..z3 and the pre-activation
// gradients dz1..dz3, dq existed in HBM only to reach the weight-gradient launch (1.1 KB written and read back per node
// row, 2.4x the algorithmic bytes of the MLP) and Dense-0's inputs h | x | agg were streamed twice.  Here a wave keeps
// ALL 68 output tiles of dW0..dW3 (F = 64: 272 accumulator registers, hence one wave per SIMD and launch_bounds(256, 1))
// and feeds them from the values it already holds.
//
// The chain keeps a node row per lane (lane (j, kg) = 4 consecutive features of row j: weights in MFMA A, rows in B);
// a weight gradient contracts over ROWS, so both its operands need the row index on the MFMA k axis (lane >> 4).  That
// transposition is one wave-private LDS round trip per 16x16 block: the lane writes its float4 at [row j][4 kg] of a
// 1 KiB block (the 64 lanes cover the block contiguously: conflict-free) and reads the dword at [q * 64 + lane], which
// is element [row 4 q + kg][feature j] -- exactly the A (or B) operand of MFMA step q, again conflict-free.  No
// workgroup barrier inside the row loop; the only cross-wave traffic is the final sum of the four accumulator sets,
// which goes through LDS in a fixed order (deterministic) and is written as this workgroup's partial-sum slab in the
// layout k_reduce_adam sums (the one k_wgrad writes).
#pragma once
#include "kernels.hpp"
#include <type_traits>

namespace v2x {

struct MlpWgLayer { int64_t layer_off, slot_stride; int n_real; RowPad pad; };
// Work split: slot s owns positions [s * slot_span, (s + 1) * slot_span) of one tile list (its tiles_per_slot 16-row tiles
// first, the rest padding); workgroup g owns positions [g * tiles_per_wg, (g + 1) * tiles_per_wg); slot_span is a multiple
// of tiles_per_wg, so a workgroup serves exactly one slot and its partial sum is slab number g - first_wg(s) < n_slabs.
// (Tried: slots packed back to back, slot_span == tiles_per_slot, with workgroups that cross a slot boundary flushing
// their partial sums and re-staging the weights in between -- 20 slots x 256 tiles over 256 CUs is then 5 tiles per
// wave for everybody instead of 6 for half of the waves.  The second staging + flush costs a crossing workgroup more
// than the round it saves (51 K cycles against 34 K), and the segment loop around the body made hipcc spill in the
// staging and flush code of every workgroup: 126 us against 117.)
struct MlpWgArgs {
  float* slab; int64_t slab_stride; MlpWgLayer l[4];
  int tiles_per_slot, slot_span, tiles_per_wg, n_slots, n_slabs;
  long long* ts;                                        // phase stamps (diagnostics) or null
};
struct MlpTrainWgArgs { MlpArgs a; MlpWgArgs w; };

template <int F>
struct MlpWgLds {
  static constexpr int KB1 = 2 * (F / 16) + 1;
  static constexpr int BLK = 256;                                   // floats of a staged 16 x 16 block
  static constexpr int PER_WAVE = (KB1 + 5 + 5) * BLK;              // z0 | K-operand scratch | N-operand scratch
  static constexpr int STAGE = (MlpLds<F>::TOTAL + 3) / 4 * 4;      // after the weight images
  static constexpr int T0 = 0, T1 = KB1 * 5, T2 = T1 + 5 * 3, T3 = T2 + 3 * 2, TILES = T3 + 2;
  static constexpr int NBIAS = (5 + 3 + 2 + 1) * 16;
  static constexpr int TOTAL = STAGE + 4 * PER_WAVE;                // floats
  static constexpr int X_SET = TILES * 256;                         // one accumulator set, [tile][lane] float4
  static constexpr int X_BIAS = 2 * X_SET;
  static_assert(X_BIAS + 4 * NBIAS <= TOTAL, "accumulator exchange must fit");
};

// same-wave LDS hand-over (lane -> lane): order the accesses for the compiler; the hardware executes one wave's LDS
// instructions in order
#define V2X_WAVE_SYNC()                                      \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    __builtin_amdgcn_sched_barrier(0);                       \
  } while (0)

// One wave per SIMD: nobody else hides an LDS round trip, and hipcc left to itself reads two weights, waits, issues two
// MFMAs.  Every MFMA run below is therefore software-pipelined BY HAND: the operands of step k+1 are requested, then the
// MFMAs of step k are issued, and sched_group_barrier spreads the requests between them (one DS instruction per PER
// MFMAs); a sched_barrier closes each step so that nothing drifts across.
#define V2X_DS_ILV(NLD, PER)                                   \
  {                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < (NLD); ++u_) {     \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       \
      __builtin_amdgcn_sched_group_barrier(0x008, (PER), 0);   \
    }                                                          \
  }                                                            \
  __builtin_amdgcn_sched_barrier(0)

// Every run takes its double buffer from the caller: PRELOADED says that buffer 0 already holds the operands of step 0
// (requested during the previous run), and `pre` -- NPRE DS instructions, typically the next run's step-0 operands -- is
// issued inside this run's LAST step, so that consecutive runs hand over without an exposed LDS round trip.
struct NoPre { __device__ __forceinline__ void operator()() const {} };

template <int NT>
__device__ __forceinline__ void fwd_weights(const float* sW, int ld, int j, int kg, int kb, float (&w)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int s = 0; s < 4; ++s) w[nt][s] = sW[(kb * 16 + 4 * kg + s) * ld + nt * 16 + j];
}
template <int NT, typename OROW>
__device__ __forceinline__ void bwd_weights(const float* sW, int ld, OROW orow, int j, int kg, int kb, float4 (&t)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) t[nt] = *reinterpret_cast<const float4*>(sW + (orow(nt) + j) * ld + kb * 16 + 4 * kg);
}
template <int NT>
__device__ __forceinline__ void wg_n_operand(const float* sN, int lane, float (&bT)[NT][4]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) bT[nt][q] = sN[nt * 256 + q * 64 + lane];
}
__device__ __forceinline__ void wg_k_operand(const float* sK, int lane, int kt, float (&aT)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) aT[q] = sK[kt * 256 + q * 64 + lane];
}

// forward layer: acc[nt] += W^T tiles x activation blocks blk[0..KB) (column reads of the weight image)
template <int KB, int NT, bool PRELOADED = false, int NPRE = 0, typename Pre = NoPre>
__device__ __forceinline__ void chain_fwd(const float* sW, int ld, int j, int kg, const f32x4* blk, f32x4 (&acc)[NT],
                                          float (&w)[2][NT][4], Pre pre = Pre()) {
  if (!PRELOADED) fwd_weights<NT>(sW, ld, j, kg, 0, w[0]);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) fwd_weights<NT>(sW, ld, j, kg, kb + 1, w[(kb + 1) & 1]);
    else pre();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = V2X_MFMA(w[kb & 1][nt][s], blk[kb][s], acc[nt]);
    if (kb + 1 < KB) { V2X_DS_ILV(2 * NT, 2); }
    else if (NPRE > 0) { V2X_DS_ILV(NPRE, (4 * NT / (NPRE > 0 ? NPRE : 1) > 0 ? 4 * NT / (NPRE > 0 ? NPRE : 1) : 1)); }
    else __builtin_amdgcn_sched_barrier(0);
  }
}

// reverse layer: acc[nt] += W tiles (rows orow(nt)..+15) x gradient blocks g[0..KB) (row reads of the image)
template <int KB, int NT, bool PRELOADED = false, int NPRE = 0, typename OROW, typename Pre = NoPre>
__device__ __forceinline__ void chain_bwd(const float* sW, int ld, OROW orow, int j, int kg, const f32x4* g, f32x4 (&acc)[NT],
                                          float4 (&t)[2][NT], Pre pre = Pre()) {
  if (!PRELOADED) bwd_weights<NT>(sW, ld, orow, j, kg, 0, t[0]);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) bwd_weights<NT>(sW, ld, orow, j, kg, kb + 1, t[(kb + 1) & 1]);
    else pre();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = V2X_MFMA(t[kb & 1][nt].x, g[kb][0], acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = V2X_MFMA(t[kb & 1][nt].y, g[kb][1], acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = V2X_MFMA(t[kb & 1][nt].z, g[kb][2], acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = V2X_MFMA(t[kb & 1][nt].w, g[kb][3], acc[nt]);
    if (kb + 1 < KB) { V2X_DS_ILV(NT, 4); }
    else if (NPRE > 0) { V2X_DS_ILV(NPRE, (4 * NT / (NPRE > 0 ? NPRE : 1) > 0 ? 4 * NT / (NPRE > 0 ? NPRE : 1) : 1)); }
    else __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[kt][nt] += K-block(kt)^T x N-block(nt) over the 16 rows of the tile; bs[nt] += column sums of the N operand
template <int KT, int NT, bool PRELOADED = false, int NPRE = 0, typename Pre = NoPre>
__device__ __forceinline__ void wg_accum(const float* sK, const float* sN, int lane, f32x4 (&acc)[KT][NT], float (&bs)[NT],
                                         float (&bT)[NT][4], float (&aT)[2][4], Pre pre = Pre()) {
  if (!PRELOADED) {
    wg_n_operand<NT>(sN, lane, bT);
    wg_k_operand(sK, lane, 0, aT[0]);
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) wg_k_operand(sK, lane, kt + 1, aT[(kt + 1) & 1]);
    else pre();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = V2X_MFMA(aT[kt & 1][q], bT[nt][q], acc[kt][nt]);
    if (kt == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bs[nt] += (bT[nt][0] + bT[nt][1]) + (bT[nt][2] + bT[nt][3]);
    }
    if (kt + 1 < KT) { V2X_DS_ILV(2, 2 * NT); }
    else if (NPRE > 0) { V2X_DS_ILV(NPRE, (4 * NT / (NPRE > 0 ? NPRE : 1) > 0 ? 4 * NT / (NPRE > 0 ? NPRE : 1) : 1)); }
    else __builtin_amdgcn_sched_barrier(0);
  }
}

// slab rows in the flat parameter order [k_real][n_real] | bias[n_real]; element (k, n) of the padded problem sits
// in tile (k >> 4, n >> 4), lane (n & 15) + 16 * ((k & 15) >> 2), register k & 3 of BOTH half sums sA, sB (layout
// [tile][lane] float4); the slab gets sA + sB.  NREAL: compile-time n_real (0: runtime)
template <int NT, int NREAL>
__device__ __forceinline__ void wg_write(const float* sA, const float* sB, const float* sBias, float* dst, const MlpWgLayer& l) {
  const int n_real = NREAL ? NREAL : l.n_real, total = l.pad.k_real * n_real;
  if constexpr (NREAL > 0 && NREAL % 4 == 0) {
    constexpr int C4 = NREAL / 4;
    const int total4 = l.pad.k_real * C4;
    for (int e = threadIdx.x; e < total4; e += 256) {
      const int rr = e / C4, col = (e - rr * C4) * 4;
      const int kp = rr < l.pad.pad_at ? rr : rr + l.pad.n_pad;
      const int i4 = kp & 15;
      const int idx = ((((kp >> 4) * NT + (col >> 4)) * 64) + (i4 >> 2) * 16 + (col & 15)) * 4 + (i4 & 3);
      *reinterpret_cast<float4*>(dst + (int64_t)rr * NREAL + col) =
          make_float4(sA[idx] + sB[idx], sA[idx + 4] + sB[idx + 4], sA[idx + 8] + sB[idx + 8], sA[idx + 12] + sB[idx + 12]);
    }
  } else {
    for (int e = threadIdx.x; e < total; e += 256) {
      const int rr = e / n_real, col = e - rr * n_real;
      const int kp = rr < l.pad.pad_at ? rr : rr + l.pad.n_pad;
      const int i4 = kp & 15;
      const int idx = ((((kp >> 4) * NT + (col >> 4)) * 64) + (i4 >> 2) * 16 + (col & 15)) * 4 + (i4 & 3);
      dst[e] = sA[idx] + sB[idx];
    }
  }
  if ((int)threadIdx.x < n_real) dst[total + threadIdx.x] = sBias[threadIdx.x];
}

// FRAG: h, agg and gha fragment-major (MlpArgs::frag_groups) -- a compile-time form: with the layout selected at run time
// the strides are no longer immediates, every access gets its own 64-bit address and the kernel spills (120 us)
// WG0 = false (round 6, small batches -- the shares of the metric's global batch): Dense-0's weight gradient is NOT taken here.
// At <= 2 tiles per wave the launch is one wave's latency chain, and dW0 is 180 of a tile's 796 MFMAs and 45 of the 68
// accumulator tiles that the end of the kernel exchanges through LDS and writes as slabs; the gated dz1 rows go to HBM instead
// (MlpArgs::dz1, 320 bytes per node row) and dW0 becomes a role of the graph layers' weight-gradient launch (k_wgrad,
// WG_KIND_DENSE0_FRAG), which at those sizes leaves a third of the chip idle.
template <int F, bool FRAG = false, bool WG0 = true>
__global__ __launch_bounds__(256, 1) void k_mlp_train_wg(MlpTrainWgArgs args) {
  using L = MlpLds<F>;
  using G = MlpWgLds<F>;
  constexpr int FB = F / 16, KB1 = 2 * FB + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MlpArgs& a = args.a;
  const MlpWgArgs& w = args.w;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, kg = lane >> 4;
  const int T = w.slot_span;
  const int wg_begin = blockIdx.x * w.tiles_per_wg, wg_end = min(wg_begin + w.tiles_per_wg, T * w.n_slots);
  const int cq = 4 * kg < a.C ? 4 * kg : 0;
  float* sZ0 = smem + G::STAGE + wv * G::PER_WAVE;
  float* sK = sZ0 + KB1 * G::BLK;
  float* sN = sK + 5 * G::BLK;
  const int wofs = j * 16 + 4 * kg;                  // where this lane's float4 of a block goes

  // diagnostics (V2X_FUSED_TS=1): shader-clock stamps of workgroup 1, one row of 64 per wave
  long long* tsp = (w.ts && blockIdx.x == 1 && lane == 0) ? w.ts + wv * 64 : nullptr;
  int tsn = 0;
  auto mark = [&]() {
    if (tsp && tsn < 64) tsp[tsn] = tsn == 0 ? (long long)wall_clock64() : (long long)__builtin_readcyclecounter();
    ++tsn;
  };
  mark();
  mark();

  const int seg_begin = wg_begin, slot = seg_begin / T, tid_l = threadIdx.x;
  const int seg_end = min(wg_end, (slot + 1) * T);
  const int tile0 = slot * T;                         // global number of the slot's first tile
  f32x4 acc0[WG0 ? KB1 : 1][5], acc1[5][3], acc2[3][2], acc3[2][1];
  float bs0[5], bs1[3], bs2[2], bs3[1];
#pragma unroll
  for (int kt = 0; kt < (WG0 ? KB1 : 1); ++kt)
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) acc0[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 5; ++kt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc1[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc2[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc3[0][0] = acc3[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) bs0[nt] = 0.f;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) bs1[nt] = 0.f;
  bs2[0] = bs2[1] = bs3[0] = 0.f;

  // (tile numbers below are local to the slot)
  auto load_in = [&](int t, MlpTrainIn<F>& in) {
    const int idx = min(t * 16 + j, a.n_idx - 1);
    const int64_t row = (int64_t)(a.idx_base + idx) * a.row_stride + slot * a.base_mul;
    in.row = row;
    const int64_t ho = FRAG ? ((int64_t)slot * a.frag_groups + min(t, a.frag_groups - 1)) * (FB * 256) + lane * 4 : row * F + 4 * kg;
    constexpr int hs = FRAG ? 256 : 16;
#pragma unroll
    for (int b = 0; b < FB; ++b) in.z0[b] = ld4(a.h + ho + b * hs);
    in.z0[FB] = ld4(a.xe + row * XE + 4 * kg);
#pragma unroll
    for (int b = 0; b < FB; ++b) in.z0[FB + 1 + b] = ld4(a.agg + ho + b * hs);
    in.y = ld4(a.y + row * a.C + cq);                    // (DQN step: the row of MlpArgs::tq = {replaced entry, action, -, -})
  };
  // `in` holds tile t on entry and tile t_next on exit: the next tile's rows are requested when only Dense-0's weight
  // gradient is left to do (180 MFMAs, ~2.5 us: enough to cover the HBM round trip) -- the chain's registers are dead
  // by then, and the 272 accumulators leave no room for a second input buffer during the chain
  float w5[2][5][4];                                   // Dense-0 forward weights: step 0 is requested by the previous tile
  auto compute = [&](int t, int t_next, MlpTrainIn<F>& in) {
    const bool valid = t * 16 + j < a.n_idx;
    const int64_t row = in.row;
    const int64_t srow = (int64_t)slot * a.srow_stride + a.idx_base + min(t * 16 + j, a.n_idx - 1);
    const auto lin = [](int nt) { return nt * 16; };
    const auto skip_xe = [](int nt) { return nt < FB ? nt * 16 : F + XE + (nt - FB) * 16; };
    float w3[2][3][4], w2[2][2][4], w1[2][1][4];         // forward weights of Dense 1..3
    float4 t2[2][2], t3[2][3], t5[2][5], t8[2][2 * FB];  // reverse weights of Dense 3..0
    float aT[2][4], b1[1][4], b2[2][4], b3[3][4], b5[5][4];
    // ================= forward (z0 is parked in LDS for Dense-0's weight gradient at the very end)
    if constexpr (WG0) {
#pragma unroll
      for (int b = 0; b < KB1; ++b) st4(sZ0 + b * G::BLK + wofs, in.z0[b]);
    }
    f32x4 z1[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) z1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bias[5];                                        // requested in the last step of each layer, like the next weights
    chain_fwd<KB1, 5, true, 11>(smem + L::W1, LD1, j, kg, in.z0, z1, w5, [&]() {
      fwd_weights<3>(smem + L::W2, LD2, j, kg, 0, w3[0]);
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) bias[nt] = ld4(smem + L::B1 + nt * 16 + 4 * kg);
    });
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) z1[nt] = relu4(z1[nt] + bias[nt]);
    mark();
    f32x4 z2[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) z2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain_fwd<5, 3, true, 7>(smem + L::W2, LD2, j, kg, z1, z2, w3, [&]() {
      fwd_weights<2>(smem + L::W3, LD3, j, kg, 0, w2[0]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) bias[nt] = ld4(smem + L::B2 + nt * 16 + 4 * kg);
    });
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) z2[nt] = relu4(z2[nt] + bias[nt]);
    f32x4 z3[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) z3[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain_fwd<3, 2, true, 4>(smem + L::W3, LD3, j, kg, z2, z3, w2, [&]() {
      fwd_weights<1>(smem + L::W4, LD4, j, kg, 0, w1[0]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bias[nt] = ld4(smem + L::B3 + nt * 16 + 4 * kg);
    });
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) z3[nt] = relu4(z3[nt] + bias[nt]);
    f32x4 qa[1];
    qa[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain_fwd<2, 1, true, 3>(smem + L::W4, LD4, j, kg, z3, qa, w1, [&]() {
      bwd_weights<2>(smem + L::W4, LD4, lin, j, kg, 0, t2[0]);
      bias[0] = ld4(smem + L::B4 + 4 * kg);
    });
    const f32x4 qv = qa[0] + bias[0];
    // targets: the row that was loaded, or (DQN step, workgroup-uniform) q itself with the taken action's entry replaced -- as
    // selects, no branch: the prefetch of the next tile must not meet a join
    const bool dqn = a.tq != nullptr;
    const int act = __float_as_int(in.y[1]);
    const float tq_v = in.y[0];
    f32x4 y4;
#pragma unroll
    for (int c = 0; c < 4; ++c) y4[c] = dqn ? (c == act ? tq_v : qv[c]) : in.y[c];
    if (valid && 4 * kg < a.C) st4(a.q + row * a.C + 4 * kg, dqn ? y4 : qv);
    // ================= Huber (delta = 1); rows past the end carry a zero gradient through everything below
    f32x4 g4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (4 * kg < a.C) {
      float ls = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float err = qv[c] - y4[c];
        const float ab = fabsf(err), quad = fminf(ab, 1.f);
        ls += 0.5f * quad * quad + (ab - quad);
        g4[c] = valid ? fminf(fmaxf(err, -1.f), 1.f) * a.inv_denom : 0.f;
      }
      if (valid) a.rowloss[srow] = ls;
    }
    mark();
    // ================= reverse.  Per layer: park the weight gradient's operands in LDS, run the DATA gradient first (it
    // needs registers and the image only), request the parked operands during its last step, then the weight gradient,
    // whose last step requests the next layer's reverse weights.
    // ---- Dense-3: d3 = W4 dq gated, dW3 += z3^T dq
#pragma unroll
    for (int b = 0; b < 2; ++b) st4(sK + b * G::BLK + wofs, z3[b]);
    st4(sN + wofs, g4);
    V2X_WAVE_SYNC();
    f32x4 d3[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) d3[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain_bwd<1, 2, true, 4>(smem + L::W4, LD4, lin, j, kg, &g4, d3, t2, [&]() { wg_n_operand<1>(sN, lane, b1); wg_k_operand(sK, lane, 0, aT[0]); });
    wg_accum<2, 1, true, 3>(sK, sN, lane, acc3, bs3, b1, aT, [&]() { bwd_weights<3>(smem + L::W3, LD3, lin, j, kg, 0, t3[0]); });
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      d3[nt] = nt * 16 + 4 * kg < H3 ? gate4(d3[nt], z3[nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- Dense-2
    V2X_WAVE_SYNC();
#pragma unroll
    for (int b = 0; b < 3; ++b) st4(sK + b * G::BLK + wofs, z2[b]);
#pragma unroll
    for (int b = 0; b < 2; ++b) st4(sN + b * G::BLK + wofs, d3[b]);
    V2X_WAVE_SYNC();
    f32x4 d2[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain_bwd<2, 3, true, 6>(smem + L::W3, LD3, lin, j, kg, d3, d2, t3, [&]() { wg_n_operand<2>(sN, lane, b2); wg_k_operand(sK, lane, 0, aT[0]); });
    wg_accum<3, 2, true, 5>(sK, sN, lane, acc2, bs2, b2, aT, [&]() { bwd_weights<5>(smem + L::W2, LD2, lin, j, kg, 0, t5[0]); });
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
      d2[nt] = nt * 16 + 4 * kg < H2 ? gate4(d2[nt], z2[nt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    mark();
    // ---- Dense-1
    V2X_WAVE_SYNC();
#pragma unroll
    for (int b = 0; b < 5; ++b) st4(sK + b * G::BLK + wofs, z1[b]);
#pragma unroll
    for (int b = 0; b < 3; ++b) st4(sN + b * G::BLK + wofs, d2[b]);
    V2X_WAVE_SYNC();
    f32x4 d1[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    chain_bwd<3, 5, true, 8>(smem + L::W2, LD2, lin, j, kg, d2, d1, t5, [&]() { wg_n_operand<3>(sN, lane, b3); wg_k_operand(sK, lane, 0, aT[0]); });
    wg_accum<5, 3, true, 2 * FB>(sK, sN, lane, acc1, bs1, b3, aT, [&]() { bwd_weights<2 * FB>(smem + L::W1, LD1, skip_xe, j, kg, 0, t8[0]); });
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) d1[nt] = gate4(d1[nt], z1[nt]);
    mark();
    // ---- Dense-0: data gradient to [dh | dagg], then dW0 += z0^T d1
    V2X_WAVE_SYNC();
    if constexpr (WG0) {
#pragma unroll
      for (int b = 0; b < 5; ++b) st4(sN + b * G::BLK + wofs, d1[b]);
    } else {                                             // dz1 rows for the Dense-0 role of the weight-gradient launch
#pragma unroll
      for (int b = 0; b < 5; ++b)
        if (valid) st4(a.dz1 + row * H1 + b * 16 + 4 * kg, d1[b]);
    }
    V2X_WAVE_SYNC();
    f32x4 o[2 * FB];
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (WG0)
      chain_bwd<5, 2 * FB, true, 12>(smem + L::W1, LD1, skip_xe, j, kg, d1, o, t8, [&]() { wg_n_operand<5>(sN, lane, b5); wg_k_operand(sZ0, lane, 0, aT[0]); });
    else
      chain_bwd<5, 2 * FB, true, 10>(smem + L::W1, LD1, skip_xe, j, kg, d1, o, t8, [&]() { fwd_weights<5>(smem + L::W1, LD1, j, kg, 0, w5[0]); });
    const int64_t go = FRAG ? ((int64_t)slot * a.frag_groups + min(t, a.frag_groups - 1)) * (2 * FB * 256) + lane * 4 : row * (2 * F) + 4 * kg;
    constexpr int gs = FRAG ? 256 : 16;
#pragma unroll
    for (int nt = 0; nt < 2 * FB; ++nt)
      if (valid) st4(a.gha + go + nt * gs, o[nt]);
    mark();
    __builtin_amdgcn_sched_barrier(0);
    load_in(t_next, in);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (WG0)
      wg_accum<KB1, 5, true, 10>(sZ0, sN, lane, acc0, bs0, b5, aT, [&]() { fwd_weights<5>(smem + L::W1, LD1, j, kg, 0, w5[0]); });
    V2X_WAVE_SYNC();
    mark();
  };

  // this wave's tiles: tile i of the workgroup belongs to wave i % 4.  (The last iteration re-requests its own tile: an
  // unconditional load keeps the loop branch-free.)
  const int first = seg_begin + ((wv - (seg_begin - wg_begin)) & 3);
  const int t_begin = first - tile0, t_end = min(seg_end - tile0, w.tiles_per_slot);
  const int nt_w = t_begin < t_end ? (t_end - t_begin + 3) >> 2 : 0;
  MlpTrainIn<F> cur;
  if (nt_w > 0) load_in(t_begin, cur);
  __builtin_amdgcn_sched_barrier(0);
  mlp_fill_lds<F>(smem, a, slot, true, tid_l);
  __syncthreads();
  mark();
  fwd_weights<5>(smem + L::W1, LD1, j, kg, 0, w5[0]);
#pragma unroll 1
  for (int k = 0; k < nt_w; ++k) compute(t_begin + 4 * k, t_begin + 4 * min(k + 1, nt_w - 1), cur);

  // ---- sum the four waves' accumulator sets, (w0 + w1) + (w2 + w3), through two LDS sets: one wave of a pair STORES its
  //      tiles, the other then adds its own in place (reads issued four tiles at a time: that is all the registers left
  //      next to 272 accumulators allow), the slab writer adds the two sets.
  //      Measured alternatives: "wave 0 stores, 1..3 read-add-write in turn" 29 us (one wave at a time, every LDS round
  //      trip exposed); read-and-add trees / butterflies 17-21 us (hipcc hoists all the reads and spills ~200 registers);
  //      LDS float atomics without return 49 us (ds_add_f32 retires a few lanes per clock).
  mark();
  __syncthreads();                                     // weight images and staging are dead from here on
  float* sBias4 = smem + G::X_BIAS;                    // [wave][11][16]
  auto fold = [&](float& v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); };
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) fold(bs0[nt]);
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) fold(bs1[nt]);
  fold(bs2[0]); fold(bs2[1]); fold(bs3[0]);
  if (lane < 16) {
    float* b = sBias4 + wv * G::NBIAS + lane;
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) b[nt * 16] = WG0 ? bs0[nt] : 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) b[(5 + nt) * 16] = bs1[nt];
    b[8 * 16] = bs2[0]; b[9 * 16] = bs2[1]; b[10 * 16] = bs3[0];
  }
  constexpr int TB = WG0 ? 0 : G::T1;                   // first tile of the exchange (WG0 = false: Dense 1..3 only, 23 tiles)
  auto each = [&](auto&& fn) {                         // fn(tile registers, tile index): indices are compile-time after unrolling
    if constexpr (WG0) {
#pragma unroll
      for (int kt = 0; kt < KB1; ++kt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) fn(acc0[kt][nt], G::T0 + kt * 5 + nt);
    }
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) fn(acc1[kt][nt], G::T1 + kt * 3 + nt);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) fn(acc2[kt][nt], G::T2 + kt * 2 + nt);
    fn(acc3[0][0], G::T3);
    fn(acc3[1][0], G::T3 + 1);
  };
  // (both waves of a pair work in both rounds: the even wave stores the first half of the tiles and adds the second,
  //  the odd wave the other way round -- a + b and b + a are the same float)
  f32x4* sSet = reinterpret_cast<f32x4*>(smem + (wv >> 1) * G::X_SET) + lane;
  constexpr int XH = TB + (G::TILES - TB) / 2;
  auto put = [&](auto LOW) {
    each([&](const f32x4& t, int ti) {
      if ((ti < XH) == decltype(LOW)::value) {
        sSet[ti * 64] = t;
        if ((ti & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  auto add = [&](auto LOW) {
    each([&](const f32x4& t, int ti) {
      if ((ti < XH) == decltype(LOW)::value) {
        sSet[ti * 64] += t;
        if ((ti & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  if ((wv & 1) == 0) put(std::true_type{}); else put(std::false_type{});
  __syncthreads();
  mark();
  if ((wv & 1) == 0) add(std::false_type{}); else add(std::true_type{});
  const float* bq = sBias4 + min((int)threadIdx.x, G::NBIAS - 1);
  const float bsum4 = (bq[0] + bq[G::NBIAS]) + (bq[2 * G::NBIAS] + bq[3 * G::NBIAS]);
  __syncthreads();
  if (threadIdx.x < G::NBIAS) sBias4[threadIdx.x] = bsum4;
  __syncthreads();
  mark();
  const float* sA = smem;
  const float* sB = smem + G::X_SET;
  const float* sBias = sBias4;
  const int first_wg = tile0 / w.tiles_per_wg;           // first workgroup with tiles of this slot
  const int my_slab = blockIdx.x - first_wg;
  float* slab = w.slab + (int64_t)my_slab * w.slab_stride;
  if constexpr (WG0)
    wg_write<5, H1>(sA + G::T0 * 256, sB + G::T0 * 256, sBias, slab + w.l[0].layer_off + slot * w.l[0].slot_stride, w.l[0]);
  wg_write<3, H2>(sA + G::T1 * 256, sB + G::T1 * 256, sBias + 5 * 16, slab + w.l[1].layer_off + slot * w.l[1].slot_stride, w.l[1]);
  wg_write<2, H3>(sA + G::T2 * 256, sB + G::T2 * 256, sBias + 8 * 16, slab + w.l[2].layer_off + slot * w.l[2].slot_stride, w.l[2]);
  wg_write<1, 0>(sA + G::T3 * 256, sB + G::T3 * 256, sBias + 10 * 16, slab + w.l[3].layer_off + slot * w.l[3].slot_stride, w.l[3]);
  mark();
  if (tsp && tsn < 64) tsp[tsn] = (long long)wall_clock64();
}

}  // namespace v2x
