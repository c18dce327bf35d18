// SPLIT-TILE variants of the graph-major fused graph layers (kernels_fused.hpp) for the SHARES of the metric's global batch.
//
// k_gnn_fwd_fused / k_gnn_bwd_fused give one workgroup 16 whole graphs: 20 node slots x 144 MFMAs per stage on the four SIMDs of
// ONE CU, a 46-49 us latency chain whatever the batch.  At the 512-graph share of an 8-GPU run (BASELINE.json: global batch 4096)
// that is 32 workgroups on a 256-CU chip.  Per-node weights (GNNLayer.build, BS_brain.py:17-42) tie an MFMA tile to ONE slot,
// so the only finer cut of a 16-graph tile is by slot: here K workgroups share a tile, member m owns the slots m, m + K, ...
// (the node update of GNNLayer.call :44-51 for those slots only: 5 slots at K = 4 instead of 20), every member keeps the
// WHOLE tile of h_s rows in its LDS (AggLayer.call :69-76 needs every source row of the graph) and the members hand each
// other their rows after every stage.
//
// The hand-over is the rollout kernel's (kernels_small.hpp, MI355X_MICROARCH.md "handoff-1to1"): rows travel as 8-byte
// {value, tag} words written and read with relaxed agent-scope atomics (global_store / global_load ... sc1), tag = 16 * epoch +
// stage code; a consumer polls the words THEMSELVES, no flag, no fence, no cache maintenance.  epoch = the tile's launch count:
// member 0 adds one when it leaves (it cannot leave before every partner has published its last rows, i.e. has read the
// count), so the epoch is the same for all members of a launch, STRICTLY larger in the next one whatever K that one runs
// with, and correct under hipGraph replay.  (The first version counted departures and divided by K: a tile that is visited by
// launches of different K -- batches of 1200, then 300 graphs -- repeated an epoch, and a member could take a stale row of the
// earlier launch for a fresh one: a 1-in-6 failure of tests/test_gpu_model.py::test_graph_cache_survives_workspace_growth.)  One slab per stage: a member that runs ahead never overwrites a row a slower
// member still needs.  Members of a tile sit on one XCD (block b runs on XCD b % 8 -- speed only, never correctness).
// Every poll is bounded (SM_POLL_CAP): a member whose partners never arrive raises FZ_ERR_XCHG and runs to its end.
//
// Measured before it was built (tools/proxy_split.sh: the unsplit kernels on 5- and 4-link graphs = the per-member work at
// K = 4 / 5): forward 20.7 / 16.1 us, backward 18.0 / 13.9 us against 49.2 / 41.5 us -- what the hand-overs may cost.
// The other candidate, 4-graph tiles on v_mfma_f32_4x4x1_16b_f32 (no hand-over at all), was measured and dropped
// (tools/mfma4bench.hip): every weight then serves 4 graphs instead of 16 and a stage streams 737 KB from L2 per workgroup,
// 7.0 us per stage at 128 and at 256 workgroups (13.5 / 26 TB/s) -- no better than the 10 us MFMA phase it replaces.
//
// What the phase stamps of the first version said (profiles/r05_split_phases_v1.txt: 33 us forward at K = 5, hand-overs of ~3 us
// each, one-slot node updates of 5-6 us waiting for weights whose first touch per XCD and step is a memory miss) shaped this one:
//   * a member owns at most 8 slots: ONE per wave, its whole weight item (36 fragments at F = 64) in registers, requested a
//     phase ahead -- no ring, no load inside an MFMA run;
//   * h_0 is not handed over at all: the embed stage is 16 MFMAs per slot, every member computes it for ALL slots;
//   * the node update of stage s + 1 starts BEFORE the partners' h_s rows are there: its [h | x | e] k-blocks need the wave's own
//     row only (5 of 9 k-blocks, in the accumulation order of the unsplit kernel: bitwise the same sums), the aggregation
//     k-blocks follow the gather; the backward computes and publishes the dagg half of a data gradient before its dh half;
//   * a unit whose words are not there yet is re-read alone (at most a few rounds), then the wave falls back to one-lane polls;
//   * an L2 warm-up of the later stages' weights (one dword per line and lane, issued and waited for at the start) was built
//     twice: with the wait in a later statement it corrupted results (hipcc re-used the destination registers while the loads
//     were in flight), self-contained it delayed the CSR chain by 2 us and bought 0.3 us later: not kept;
//   * the number of graph layers L is a template parameter (1..3) and the stage loop is unrolled: a weight request that is in
//     flight across a loop's back edge makes hipcc's wait-count pass wait for vmcnt(0) at its first use -- and with it for
//     every younger prefetch (the second version of these kernels: 40 us where the first took 37.6).
//
// Edge form only (the general edge-index aggregation: what bench.py's `value` runs); arithmetic order is that of the unsplit
// kernels, so the results are bitwise theirs.
#pragma once
#include "kernels_fused.hpp"
#include "kernels_small.hpp"

namespace v2x {

constexpr int FZ_ERR_XCHG = 1 << 9;                  // flag word: a split-tile member timed out waiting for its partners' rows

struct FzXchg {
  unsigned long long* buf;                           // [slab][cap_tiles][N][FB][64 lanes][4] tagged words
  unsigned long long* sync;                          // [cap_tiles] launches that visited the tile (64-bit: never wraps), from a process-wide base
  int cap_tiles;                                     // tiles per slab
  int K;                                             // members per tile
};

// block -> (tile, member): the K members of a tile on one XCD
struct FzSplitId { int tile, m, n_tiles; };
__device__ __forceinline__ FzSplitId fz_split_id(int n_graphs, int K) {
  FzSplitId s;
  const int b = blockIdx.x, xcd = b & 7, r = b >> 3;
  s.n_tiles = (n_graphs + FZ_TG - 1) / FZ_TG;
  s.tile = (r / K) * 8 + xcd;
  s.m = r % K;
  return s;
}

// the j-th slot that member m does NOT own (ascending); may be >= N in the last block of K
__device__ __forceinline__ int fz_foreign_slot(int j, int K, int m) {
  const int b = j / (K - 1), r = j - b * (K - 1);
  return b * K + (r < m ? r : r + 1);
}

// Two tagged words per 16-byte access: {value, tag, value, tag} written with ONE global_store_dwordx4 sc0 sc1 (a 16-byte
// write-through store costs what a plain one does; 8-byte ones are a fabric write each, 2.7x the time per byte) and read with
// ONE global_load_dwordx4 sc1; each 8-byte half is single-copy atomic (MI355X_MICROARCH.md: observed untorn for 16-byte sc1
// halves), and every half carries its own tag, so a torn PAIR is still two valid words or a retry.  hipcc has no builtin for
// either; as inline asm they are invisible to its wait-count pass, which only makes its own waits conservative (the counter is
// in order); every load is waited for inside the asm statement that issues it.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fz_publish(unsigned long long* slab_tile, int slot, int nt, int FB, int lane, f32x4 v, unsigned tag) {
  unsigned long long* p = slab_tile + ((int64_t)(slot * FB + nt) * 64 + lane) * 4;
  const u32x4 lo = {__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag}, hi = {__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag};
  // (s_nop: a store of more than 64 bits followed by a VALU write of its data registers needs wait states the compiler's hazard
  //  recogniser cannot place for an instruction it does not see)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(lo), "v"(hi) : "memory");
}
// A unit = 256 tagged words = two 16-byte loads per lane.  Load AND wait are ONE asm statement: to the compiler an asm's outputs
// exist when the statement ends, so it may copy or spill them at once -- with the wait in a later statement that would be a copy
// of registers the data has not reached yet (seen: prefetches with discarded results, whose registers hipcc spilled and re-used
// while the loads were in flight, corrupted the backward at L = 3).
__device__ __forceinline__ void fz_load_unit_wait(const unsigned long long* p, u32x4& lo, u32x4& hi) {
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
}
__device__ __forceinline__ void fz_load_units8_wait(const unsigned long long* const (&p)[8], u32x4 (&t)[8][2]) {
  asm volatile(
      "global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %17, off sc1\n\tglobal_load_dwordx4 %3, %17, off offset:16 sc1\n\t"
      "global_load_dwordx4 %4, %18, off sc1\n\tglobal_load_dwordx4 %5, %18, off offset:16 sc1\n\t"
      "global_load_dwordx4 %6, %19, off sc1\n\tglobal_load_dwordx4 %7, %19, off offset:16 sc1\n\t"
      "global_load_dwordx4 %8, %20, off sc1\n\tglobal_load_dwordx4 %9, %20, off offset:16 sc1\n\t"
      "global_load_dwordx4 %10, %21, off sc1\n\tglobal_load_dwordx4 %11, %21, off offset:16 sc1\n\t"
      "global_load_dwordx4 %12, %22, off sc1\n\tglobal_load_dwordx4 %13, %22, off offset:16 sc1\n\t"
      "global_load_dwordx4 %14, %23, off sc1\n\tglobal_load_dwordx4 %15, %23, off offset:16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[2][0]), "=&v"(t[2][1]), "=&v"(t[3][0]), "=&v"(t[3][1]),
        "=&v"(t[4][0]), "=&v"(t[4][1]), "=&v"(t[5][0]), "=&v"(t[5][1]), "=&v"(t[6][0]), "=&v"(t[6][1]), "=&v"(t[7][0]), "=&v"(t[7][1])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}

// The partners' rows of one stage into the LDS tile: all waves, 8 units (slot, n-tile) of 256 words per wave and round -- at
// most 64 units over 8 waves, i.e. ONE round: a wave's whole share is requested before the first word is looked at (the words
// were written through to memory, a read is a trip to the fabric).  A unit whose tags are not yet this stage's is requested
// again, alone; after FZ_FAST_ROUNDS such rounds the wave waits CHEAPLY -- lane uu polls the last word of unit uu (one load of
// <= 8 lanes per round, s_sleep in between: a wave that keeps re-reading its whole batch is 16 TB/s of polling over the chip
// and slows the very stores it waits for -- measured, the first version of this function) -- and then reads what is missing.
constexpr int FZ_FAST_ROUNDS = 3;
template <int FB, int ROWF>
__device__ __forceinline__ void fz_collect(const unsigned long long* slab_tile, float* tileb, int SUB, int N, int K, int m, unsigned tag,
                                           int lane, int wv, int jc, int kg, int* err) {
  constexpr int U = 8;
  const int units = (K - 1) * ((N + K - 1) / K) * FB;
  for (int u0 = wv; u0 < units; u0 += FZ_WAVES * U) {
    int slot[U], nt[U];
    const unsigned long long* p[U];
    const unsigned long long* sentinel = slab_tile;
    unsigned pend = 0u;
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
      const int u = min(u0 + FZ_WAVES * uu, units - 1);
      nt[uu] = u % FB;
      slot[uu] = fz_foreign_slot(u / FB, K, m);
      const unsigned long long* ub = slab_tile + (int64_t)(min(slot[uu], N - 1) * FB + nt[uu]) * 256;
      p[uu] = ub + lane * 4;
      if (lane == uu) sentinel = ub + 255;
      if (slot[uu] < N && u0 + FZ_WAVES * uu < units) pend |= 1u << uu;       // (wave-uniform)
    }
    const unsigned want = pend;
    int rounds = 0;
    u32x4 t[U][2];
    fz_load_units8_wait(p, t);                       // (units beyond the wave's share re-read a valid address; never looked at)
    for (;;) {
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        const bool ok = t[uu][0][1] == tag && t[uu][0][3] == tag && t[uu][1][1] == tag && t[uu][1][3] == tag;
        if (((pend >> uu) & 1u) && __all(ok)) pend &= ~(1u << uu);
      }
      if (!pend) break;
      if (++rounds >= SM_POLL_CAP) { if (err && lane == 0) atomicOr(err, FZ_ERR_XCHG); break; }
      if (rounds >= FZ_FAST_ROUNDS) {                // the missing units' last words, one lane each
        for (;;) {
          const bool there = !(lane < U && ((pend >> lane) & 1u)) || (unsigned)(ld_tagged(sentinel) >> 32) == tag;
          if (__all(there) || ++rounds >= SM_POLL_CAP) break;
          __builtin_amdgcn_s_sleep(8);
        }
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int uu = 0; uu < U; ++uu)
        if ((pend >> uu) & 1u) fz_load_unit_wait(p[uu], t[uu][0], t[uu][1]);
    }
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
      if ((want >> uu) & 1u) {
        const f32x4 v = {__uint_as_float(t[uu][0][0]), __uint_as_float(t[uu][0][2]), __uint_as_float(t[uu][1][0]), __uint_as_float(t[uu][1][2])};
        st4(tileb + kg * SUB + (slot[uu] * FZ_TG + jc) * ROWF + nt[uu] * 4, v);
      }
    }
  }
}

// phase time stamps of one workgroup (100 MHz wall clock; measurement builds only, V2X_FUSED_TS=1: a stamp is a store, and a
// store through a generic pointer -- or a conditional one the wait-count pass has to merge -- costs every later wait its count)
template <bool TS>
struct FzStampR {
  long long* p; int n;
  __device__ __forceinline__ FzStampR(long long* base, int wv, int lane, int block) : p(nullptr), n(0) {
    if (TS && base && (int)blockIdx.x == block && lane == 0) p = base + wv * 64;
  }
  __device__ __forceinline__ void mark() {
    if (TS) {
      if (p && n < 64) *(__attribute__((address_space(1))) long long*)(p + n) = wall_clock64();
      ++n;
    }
  }
};

// "These registers must hold their values HERE": an empty asm per register makes hipcc place its wait for a weight request at
// this point -- chosen so that no store is in flight yet.  Loads and stores share the one in-order vmcnt counter but are
// acknowledged independently, so a load that is waited for AFTER a store was issued costs that store's acknowledgement as well
// (vmcnt(0)); the write-through publishes of these kernels are acknowledged late (phase stamps: 1.3 us in front of the next
// stage's own-row k-blocks, 2.4 us in front of stage 1).
template <int NV>
__device__ __forceinline__ void fz_consume(const f32x4 (&w)[NV]) {
#pragma unroll
  for (int u = 0; u < NV; ++u) asm volatile("" ::"v"(w[u]));
}

struct FzCtxS {
  float* sH; int* sRp; unsigned char* sCol; unsigned* sC; float* sBias;
  int N, SUB, lane, wv, jc, kg, g0, tile, m, K, n_tiles;
};

// ---------------------------------------------------------------------------------------------------------------
// forward.  HAS: this wave owns a slot (k = m + K * wave); the others only help with the embed stage and the hand-overs.
// Slabs: stage s = 1..L in slab s - 1 (tag code s); the backward's dagg_{s-1}, s = L..1, in slab L + s - 1 (tag code L + s).
// ---------------------------------------------------------------------------------------------------------------
constexpr int FZ_EMB = 4;                            // embed slots per wave: N <= 32

template <int F, int L, bool HAS, bool TS>
__device__ __forceinline__ void fused_fwd_split_body(const FusedFwdArgs& a, const FzXchg& xg, const FzCtxS& x, const int nrows, const int r_begin) {
  using P = FzPack<F>;
  constexpr int FB = P::FB, KB = P::KB, ROWF = P::ROWF;
  constexpr int NPRE = (FB + 1) * FB, NPOST = FB * FB;           // fragments of the k-blocks [0, FB] = [h | x e] and (FB, 2 FB] = agg
  const int N = x.N, lane = x.lane, wv = x.wv, kg = x.kg, jc = x.jc, K = x.K;
  const int k = HAS ? x.m + K * wv : 0;                           // the wave's slot
  FzStampR<TS> ts(a.ts, wv, lane, 8);
  ts.mark();                                                     // 0: start

  // this launch's epoch of the tile = its launch count (member 0 advances it when it leaves: not before everybody has published)
  const unsigned epoch16 = 16u * (unsigned)(__hip_atomic_load(xg.sync + x.tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x0fffffffull);
  auto slab_tile = [&](int slab) { return xg.buf + ((int64_t)slab * xg.cap_tiles + x.tile) * ((int64_t)N * FB * 256); };

  const int64_t rowk = (int64_t)(x.g0 + jc) * N + k;
  auto item = [&](int s) -> const float* {
    return a.pk + (int64_t)a.S * P::FWD0 + ((int64_t)(s - 1) * a.S + (a.S == 1 ? 0 : k)) * P::FWD;
  };
  auto load_pre = [&](f32x4 (&wpre)[NPRE], int s) {
    gvec_p wp = (gvec_p)item(s) + lane;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) wpre[u] = wp[u * 64];
  };
  auto load_post = [&](f32x4 (&wpost)[NPOST], int s) {
    gvec_p wp = (gvec_p)item(s) + lane + NPRE * 64;
#pragma unroll
    for (int u = 0; u < NPOST; ++u) wpost[u] = wp[u * 64];
  };
  auto load_bias = [&](f32x4 (&bias)[FB], int s) {
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) bias[nt] = ldg4(item(s) + KB * FB * 256 + nt * 16 + 4 * kg);
  };

  // ---- requests in the order of their urgency: CSR slice, the embed operands, the first weights
  FzCsrEarly csr;
  csr_issue(csr, a.row_ptr, a.col_idx, r_begin, nrows, x.g0, a.edges_cap, a.n_edges);
  // embed stage for ALL slots q = wave, wave + 8, ...: h_0 never travels between the members.  The embed biases of all slots
  // go through LDS (one float4 per thread: N * F / 4 <= 512) instead of 16 registers per slot and lane.
  f32x4 ebias = {0.f, 0.f, 0.f, 0.f};
  {
    const int t = min((int)threadIdx.x, N * F / 4 - 1), q = t / (F / 4);
    ebias = ldg4(a.pk + (int64_t)(a.S == 1 ? 0 : q) * P::FWD0 + FB * 256 + (t - q * (F / 4)) * 4);
  }
  f32x4 exe[FZ_EMB], ew[FZ_EMB][FB];
#pragma unroll
  for (int e = 0; e < FZ_EMB; ++e) {
    const int q = min(wv + FZ_WAVES * e, N - 1);
    const float* wb = a.pk + (int64_t)(a.S == 1 ? 0 : q) * P::FWD0;
    exe[e] = ldg4(a.xe + ((int64_t)(x.g0 + jc) * N + q) * XE + 4 * kg);
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) ew[e][nt] = ((gvec_p)wb + lane)[nt * 64];
  }
  f32x4 xev = {0.f, 0.f, 0.f, 0.f};
  if (HAS) xev = ldg4(a.xe + rowk * XE + 4 * kg);
  f32x4 wpre1[NPRE];
  if (HAS && L >= 1) load_pre(wpre1, 1);
  const int e_begin = a.row_ptr[r_begin], nedges = a.row_ptr[r_begin + nrows] - e_begin;
  if (nedges > a.edges_cap || nedges < 0) {                      // workgroup-uniform, before any barrier.  (The partners see the
    if (threadIdx.x == 0 && a.err) atomicOr(a.err, 1);           //  same slice and leave as well: nobody waits for anybody.)
    if (threadIdx.x == 0 && x.m == 0) __hip_atomic_fetch_add(xg.sync + x.tile, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  csr_commit(csr, a.row_ptr, a.col_idx, x.sRp, x.sCol, N, r_begin, nrows, e_begin, nedges);
  if ((int)threadIdx.x < N * F / 4) st4(x.sBias + threadIdx.x * 4, ebias);
  fz_barrier();                                                  // (LDS only: the weight requests stay in flight)
  for (int r = threadIdx.x; r < FZ_TG * N; r += FZ_THREADS) {     // the CSR rows of the tile as bit sets
    unsigned nb = 0u;
    if (r < nrows)
      for (int e = x.sRp[r]; e < x.sRp[r + 1]; ++e) nb |= 1u << x.sCol[e];
    x.sC[r] = nb;
    if (a.nbmask && r < nrows && x.m == 0) a.nbmask[r_begin + r] = nb;         // for the backward; one member writes them
  }
  ts.mark();                                                     // 1: CSR slice in LDS, sets built

  typedef const __attribute__((address_space(4))) uint64_t* CQ;
  CQ kq = (CQ)__builtin_amdgcn_kernarg_segment_ptr();
  auto hptr = [&](int s) { return reinterpret_cast<float*>(kq[offsetof(FusedFwdArgs, h) / 8 + s]); };
  auto aptr = [&](int s) { return reinterpret_cast<float*>(kq[offsetof(FusedFwdArgs, a) / 8 + s]); };
  auto gptr = [&](int s) { return reinterpret_cast<unsigned short*>(kq[offsetof(FusedFwdArgs, gate) / 8 + s]); };
  const int gate_lane = x.tile * 64 + lane, gate_stride = x.n_tiles * 64;       // the TILE's place, not the block's
  float* myrow = x.sH + kg * x.SUB + jc * ROWF;
  {
    float* hp = hptr(0);
    unsigned short* gp = gptr(0);
#pragma unroll
    for (int e = 0; e < FZ_EMB; ++e) {
      const int q = wv + FZ_WAVES * e;
      if (q < N) {                                               // (wave-uniform)
        f32x4 acc[FB];
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int nt = 0; nt < FB; ++nt) acc[nt] = V2X_MFMA(ew[e][nt][s4], exe[e][s4], acc[nt]);
        const bool own = q % K == x.m;
        unsigned gb = 0u;
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) {
          const f32x4 v = relu4(acc[nt] + ld4(x.sBias + q * F + nt * 16 + 4 * kg));
          st4(myrow + q * FZ_TG * ROWF + nt * 4, v);
          if (own) stg4(hp + ((int64_t)(x.g0 + jc) * N + q) * F + nt * 16 + 4 * kg, v);
          gb |= gate_bits4(v) << (4 * nt);
        }
        if (own) gp[q * gate_stride + gate_lane] = (unsigned short)gb;
      }
    }
  }
  // Weight requests from here on are placed so that a conservative wait (vmcnt(0): what hipcc falls back to behind any control
  // flow it cannot count through) costs nothing: a request goes out either right in front of a gather (1.3 us of LDS work; the
  // region is L2-warm by then) or right in front of a hand-over, whose own round trip to the fabric waits for everything.
  f32x4 wpost[NPOST], bias[FB];
  ts.mark();                                                     // 2: embed done (all slots)
  fz_barrier();
  ts.mark();                                                     // 3: barrier: the h_0 tile is complete

  // the node update of stage s in two parts (same accumulation order as the unsplit kernel: k-blocks 0 .. 2 FB)
  f32x4 acc[FB];
  auto pre = [&](const f32x4 (&wpre)[NPRE], const f32x4 (&hb)[FB]) {      // [h | x e] k-blocks: the wave's own row only
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb <= FB; ++kb) {
      const f32x4 bv = kb < FB ? hb[kb < FB ? kb : 0] : xev;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) acc[nt] = V2X_MFMA(wpre[kb * FB + nt][s4], bv[s4], acc[nt]);
    }
  };
  auto post = [&](const f32x4 (&ag)[FB]) {                       // aggregation k-blocks
#pragma unroll
    for (int kb = 0; kb < FB; ++kb)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) acc[nt] = V2X_MFMA(wpost[kb * FB + nt][s4], ag[kb][s4], acc[nt]);
  };
  // Edge form of AggLayer.call (BS_brain.py:69-76) for the wave's slot: ascending sources, the tile's rows walked once
  // (multiply by the 0 / 1 membership bit: the unsplit kernel's gather_all, bitwise its sums)
  auto gather = [&](f32x4 (&ag)[FB]) {
    const unsigned msk = x.sC[jc * N + k];
#pragma unroll
    for (int kb = 0; kb < FB; ++kb) ag[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (__any(2 * __popc(msk) < N)) {      // degree-aware (see the unsplit kernel's gather_all): sparse lanes walk their sources only
      unsigned walk = 2 * __popc(msk) < N ? msk : (N >= 32 ? 0xffffffffu : (1u << N) - 1u);
      while (walk) {
        const int p = __builtin_ctz(walk);
        walk &= walk - 1;
        const float* bp = myrow + p * (FZ_TG * ROWF);
        const float f = (float)((msk >> p) & 1u);
#pragma unroll
        for (int kb = 0; kb < FB; ++kb) ag[kb] += ld4(bp + kb * 4) * f;
      }
      return;
    }
    for (int p = 0; p < N; ++p) {
      const float* bp = myrow + p * (FZ_TG * ROWF);
      const float f = (float)((msk >> p) & 1u);
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) ag[kb] += ld4(bp + kb * 4) * f;
    }
  };

  if (HAS && L >= 1) {
    f32x4 hb[FB];
#pragma unroll
    for (int kb = 0; kb < FB; ++kb) hb[kb] = ld4(myrow + k * FZ_TG * ROWF + kb * 4);
    pre(wpre1, hb);
    __builtin_amdgcn_sched_barrier(0);     // (the stage's requests below must not be hoisted in front of these MFMAs' waits)
  }
  // ---- stages 1..L (unrolled: L is a template parameter)
#pragma unroll
  for (int s = 1; s <= L; ++s) {
    float* hp = hptr(s);
    float* ap = aptr(s - 1);
    unsigned short* gp = gptr(s);
    unsigned long long* out = slab_tile(s - 1);
    const unsigned tag = epoch16 + (unsigned)s;
    f32x4 v[FB], wpre[NPRE];
    if (HAS) {
      f32x4 ag[FB];
      if (s == 1) { load_post(wpost, 1); load_bias(bias, 1); }
      if (s < L) load_pre(wpre, s + 1);    // (used after this stage's barrier)
      gather(ag);
      fz_consume(wpost); fz_consume(bias);   // the waits for this stage's (and the next own part's) weights: before the first store
      if (s < L) fz_consume(wpre);
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) stg4(ap + rowk * F + kb * 16 + 4 * kg, ag[kb]);
      ts.mark();                                                 // stage: gather done
      post(ag);
      ts.mark();                                                 // stage: aggregation k-blocks issued
      unsigned gb = 0u;
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) {
        v[nt] = acc[nt] + bias[nt];
        if (s < L) v[nt] = relu4(v[nt]);
        gb |= gate_bits4(v[nt]) << (4 * nt);
      }
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) fz_publish(out, k, nt, FB, lane, v[nt], tag);         // the partners wait for these
      ts.mark();                                                 // stage: published
      if (a.frag_out && s == L) {
        float* hf = hp + ((int64_t)k * x.n_tiles + x.tile) * (FB * 256) + lane * 4;
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) stg4(hf + nt * 256, v[nt]);
      } else {
        float* hr = hp + rowk * F + 4 * kg;
#pragma unroll
        for (int nt = 0; nt < FB; ++nt) stg4(hr + nt * 16, v[nt]);
      }
      gp[k * gate_stride + gate_lane] = (unsigned short)gb;
    } else {
      ts.mark(); ts.mark(); ts.mark();
    }
    ts.mark();                                                   // stage: node update done and published
    fz_barrier();                          // every wave is done reading the h_{s-1} tile
    ts.mark();                                                   // stage: barrier
    if (HAS) {
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(myrow + k * FZ_TG * ROWF + nt * 4, v[nt]);
      if (s < L) {
        pre(wpre, v);                      // the next stage's own-row k-blocks run while the partners' rows travel
        load_post(wpost, s + 1);           // (land during the hand-over)
        load_bias(bias, s + 1);
      }
    }
    ts.mark();                                                   // stage: own part of the next update done
    fz_collect<FB, ROWF>(out, x.sH, x.SUB, N, K, x.m, tag, lane, wv, jc, kg, a.err);          // the partners' h_s rows
    ts.mark();                                                   // stage: partners' rows in the tile
    fz_barrier();
    ts.mark();                                                   // stage: barrier
  }

  // ---- a_L = Agg(h_L) for the decision MLP
  if (HAS) {
    float* ap = aptr(L);
    f32x4 agl[FB];
    gather(agl);
    if (a.frag_out) {
      float* af = ap + ((int64_t)k * x.n_tiles + x.tile) * (FB * 256) + lane * 4;
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) stg4(af + kb * 256, agl[kb]);
    } else {
      float* ar = ap + rowk * F + 4 * kg;
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) stg4(ar + kb * 16, agl[kb]);
    }
  }
  ts.mark();                                                     // end (stores issued)
  // member 0: the next launch's epoch; nobody waits for the add
  if (threadIdx.x == 0 && x.m == 0) __hip_atomic_fetch_add(xg.sync + x.tile, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a member owns at most 8 slots (host: ceil(N / K) <= 8): one per wave
template <int F, int L, bool TS = false>
__global__ __launch_bounds__(FZ_THREADS, 2) void k_gnn_fwd_split(FusedFwdArgs a, FzXchg xg) {
  using P = FzPack<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FzSplitId id = fz_split_id(a.n_graphs, xg.K);
  if (id.tile >= id.n_tiles) return;
  FzCtxS x;
  x.N = a.N; x.K = xg.K; x.tile = id.tile; x.m = id.m; x.n_tiles = id.n_tiles;
  x.SUB = a.N * FZ_TG * P::ROWF;
  x.sH = smem;
  x.sBias = x.sH + 4 * x.SUB;                                    // [N][F] embed biases
  x.sRp = reinterpret_cast<int*>(x.sBias + a.N * F);
  x.sC = reinterpret_cast<unsigned*>(x.sRp + FZ_TG * a.N + 1);
  x.sCol = reinterpret_cast<unsigned char*>(x.sC + FZ_TG * a.N);
  x.lane = threadIdx.x & 63;
  x.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  x.kg = x.lane >> 4;
  x.g0 = id.tile * FZ_TG;
  const int ng = min(FZ_TG, a.n_graphs - x.g0);
  x.jc = min(x.lane & 15, ng - 1);
  const int r_begin = x.g0 * a.N, nrows = ng * a.N;
  if (id.m + xg.K * x.wv < a.N) fused_fwd_split_body<F, L, true, TS>(a, xg, x, nrows, r_begin);
  else fused_fwd_split_body<F, L, false, TS>(a, xg, x, nrows, r_begin);
}

// ---------------------------------------------------------------------------------------------------------------
// backward: the rows' in-neighbour sets come from the forward (FusedFwdArgs::nbmask), the dagg_L rows of the partners'
// slots straight from gha, the dagg_{s-1} rows of every later stage through the exchange
// ---------------------------------------------------------------------------------------------------------------
struct FzCtxBS {
  float* sD; int* sRp; unsigned* sM;
  int N, SUB, lane, wv, jc, kg, g0, tile, m, K, n_tiles;
};

template <int F, int L, bool HAS, bool TS>
__device__ __forceinline__ void fused_bwd_split_body(const FusedBwdArgs& a, const FzXchg& xg, const FzCtxBS& x, const int nrows, const int r_begin) {
  using P = FzPack<F>;
  constexpr int FB = P::FB, ROWF = P::ROWF;
  const int N = x.N, lane = x.lane, wv = x.wv, kg = x.kg, jc = x.jc, K = x.K;
  const int k = HAS ? x.m + K * wv : 0;
  const unsigned epoch16 = 16u * (unsigned)(__hip_atomic_load(xg.sync + x.tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x0fffffffull);
  auto slab_tile = [&](int slab) { return xg.buf + ((int64_t)slab * xg.cap_tiles + x.tile) * ((int64_t)N * FB * 256); };
  FzStampR<TS> ts(a.ts ? a.ts + 512 : nullptr, wv, lane, 8);
  ts.mark();                                                     // 0: start

  const int64_t rowk = (int64_t)(x.g0 + jc) * N + k;
  // weights of (stage s, slot k): fragments [k-block][n-tile], n-tiles 0 .. FB - 1 = dh columns, FB .. 2 FB - 1 = dagg columns
  auto item = [&](int s) -> const float* { return a.pk + ((int64_t)(s - 1) * a.S + (a.S == 1 ? 0 : k)) * P::BWD; };
  auto load_half = [&](f32x4 (&w)[FB * FB], int s, int half) {
    gvec_p wp = (gvec_p)item(s) + lane + half * FB * 64;
#pragma unroll
    for (int kb = 0; kb < FB; ++kb)
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) w[kb * FB + nt] = wp[(kb * 2 * FB + nt) * 64];
  };
  float* myrow = x.sD + kg * x.SUB + jc * ROWF;
  const unsigned nbv = a.nbmask[r_begin + min((int)threadIdx.x, nrows - 1)];
  f32x4 dg[FB], dhk[FB];
  const bool frg = a.frag_gha != 0;
  const int gst = frg ? 256 : 16;
  auto gha_off = [&](int slot) -> int64_t {
    return frg ? ((int64_t)slot * x.n_tiles + x.tile) * (2 * FB * 256) + lane * 4 : ((int64_t)(x.g0 + jc) * N + slot) * (2 * F) + 4 * kg;
  };
  if (HAS) {
    const int64_t go = gha_off(k);
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) {
      dg[nt] = ldg4(a.gha + go + (FB + nt) * gst);
      dhk[nt] = ldg4(a.gha + go + nt * gst);
    }
  }
  if ((int)threadIdx.x < FZ_TG * N) x.sRp[threadIdx.x] = (int)nbv;
  {                                        // the partners' dagg_L rows, straight from gha (the MLP launch wrote them)
    const int units = (K - 1) * ((N + K - 1) / K) * FB;
    for (int u0 = wv; u0 < units; u0 += FZ_WAVES * 8) {
      f32x4 r[8];
      int so[8], no[8];
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) {
        const int u = min(u0 + FZ_WAVES * uu, units - 1);
        no[uu] = u % FB;
        so[uu] = fz_foreign_slot(u / FB, K, x.m);
        r[uu] = ldg4(a.gha + gha_off(min(so[uu], N - 1)) + (FB + no[uu]) * gst);
      }
#pragma unroll
      for (int uu = 0; uu < 8; ++uu)
        if (so[uu] < N && u0 + FZ_WAVES * uu < units) st4(myrow + so[uu] * FZ_TG * ROWF + no[uu] * 4, r[uu]);
    }
  }
  f32x4 wa[FB * FB], wh[FB * FB];
  if (HAS && L >= 1) { load_half(wa, L, 1); load_half(wh, L, 0); }
  if (HAS) {
#pragma unroll
    for (int nt = 0; nt < FB; ++nt) st4(myrow + k * FZ_TG * ROWF + nt * 4, dg[nt]);
  }
  fz_barrier();
  for (int r = threadIdx.x; r < nrows; r += FZ_THREADS) {         // transposed adjacency: bit q of sM[j*N + p] = edge p -> q
    const int jj = r / N, p = r - jj * N;
    unsigned ns = 0u;
    for (int q = 0; q < N; ++q) ns |= (((unsigned)x.sRp[jj * N + q] >> p) & 1u) << q;
    x.sM[r] = ns;
  }
  fz_barrier();
  if (HAS && L >= 1) { fz_consume(wa); fz_consume(wh); }
  ts.mark();                                                     // 1: tile + masks ready

  typedef const __attribute__((address_space(4))) uint64_t* CQ;
  CQ kq = (CQ)__builtin_amdgcn_kernarg_segment_ptr();
  auto gptr = [&](int s) { return reinterpret_cast<const unsigned short*>(kq[offsetof(FusedBwdArgs, gate) / 8 + s]); };
  const int gate_lane = x.tile * 64 + lane, gate_stride = x.n_tiles * 64;
  auto dptr = [&](int s) { return reinterpret_cast<float*>(kq[offsetof(FusedBwdArgs, dpre) / 8 + s]); };

#pragma unroll
  for (int s = L; s >= 0; --s) {
    float* dp = dptr(s);
    f32x4 dpre[FB];
    if (HAS) {
      unsigned gb = 0u;
      if (s < L) gb = gptr(s)[k * gate_stride + gate_lane];
      // transposed gather, ascending destinations (k_agg_small<true> order): the dagg tile's rows walked once
      const unsigned msk = x.sM[jc * N + k];
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) dpre[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (__any(2 * __popc(msk) < N)) {    // degree-aware
        unsigned walk = 2 * __popc(msk) < N ? msk : (N >= 32 ? 0xffffffffu : (1u << N) - 1u);
        while (walk) {
          const int q = __builtin_ctz(walk);
          walk &= walk - 1;
          const float* bq = myrow + q * (FZ_TG * ROWF);
          const float f = (float)((msk >> q) & 1u);
#pragma unroll
          for (int kb = 0; kb < FB; ++kb) dpre[kb] += ld4(bq + kb * 4) * f;
        }
      } else {
        for (int q = 0; q < N; ++q) {
          const float* bq = myrow + q * (FZ_TG * ROWF);
          const float f = (float)((msk >> q) & 1u);
#pragma unroll
          for (int kb = 0; kb < FB; ++kb) dpre[kb] += ld4(bq + kb * 4) * f;
        }
      }
#pragma unroll
      for (int kb = 0; kb < FB; ++kb) {
        f32x4 t = dpre[kb] + dhk[kb];
        if (s < L) t = gate_apply4(t, gb >> (4 * kb));
        stg4(dp + rowk * F + kb * 16 + 4 * kg, t);
        dpre[kb] = t;
      }
    }
    ts.mark();                                                   // stage: gather done, dpre_s stored
    if (s == 0) break;
    unsigned long long* out = slab_tile(L + s - 1);
    const unsigned tag = epoch16 + (unsigned)(L + s);
    if (HAS) {
      // the dagg half of the data gradient first: the partners wait for it; the dh half stays in this wave
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) dg[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < FB; ++kb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int nt = 0; nt < FB; ++nt) dg[nt] = V2X_MFMA(wa[kb * FB + nt][s4], dpre[kb][s4], dg[nt]);
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) fz_publish(out, k, nt, FB, lane, dg[nt], tag);
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) dhk[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < FB; ++kb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int nt = 0; nt < FB; ++nt) dhk[nt] = V2X_MFMA(wh[kb * FB + nt][s4], dpre[kb][s4], dhk[nt]);
    }
    ts.mark();                                                   // stage: data gradient done (dagg half published half-way)
    fz_barrier();                       // all gathers from the dagg_s tile are done
    ts.mark();                                                   // stage: barrier
    if (HAS) {
#pragma unroll
      for (int nt = 0; nt < FB; ++nt) st4(myrow + k * FZ_TG * ROWF + nt * 4, dg[nt]);
      if (s >= 2) { load_half(wa, s - 1, 1); load_half(wh, s - 1, 0); }        // (land during the hand-over, whose wait covers them)
    }
    fz_collect<FB, ROWF>(out, x.sD, x.SUB, N, K, x.m, tag, lane, wv, jc, kg, a.err);
    if (HAS && s >= 2) { fz_consume(wa); fz_consume(wh); }       // (landed during the hand-over: hipcc's wait goes here, in front of the next stores)
    ts.mark();                                                   // stage: partners' rows in the tile
    fz_barrier();
    ts.mark();                                                   // stage: barrier
  }
  ts.mark();                                                     // end
  if (threadIdx.x == 0 && x.m == 0) __hip_atomic_fetch_add(xg.sync + x.tile, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int F, int L, bool TS = false>
__global__ __launch_bounds__(FZ_THREADS, 2) void k_gnn_bwd_split(FusedBwdArgs a, FzXchg xg) {
  using P = FzPack<F>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FzSplitId id = fz_split_id(a.n_graphs, xg.K);
  if (id.tile >= id.n_tiles) return;
  FzCtxBS x;
  x.N = a.N; x.K = xg.K; x.tile = id.tile; x.m = id.m; x.n_tiles = id.n_tiles;
  x.SUB = a.N * FZ_TG * P::ROWF;
  x.sD = smem;
  x.sRp = reinterpret_cast<int*>(x.sD + 4 * x.SUB);
  x.sM = reinterpret_cast<unsigned*>(x.sRp + FZ_TG * a.N + 1);
  x.lane = threadIdx.x & 63;
  x.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  x.kg = x.lane >> 4;
  x.g0 = id.tile * FZ_TG;
  const int ng = min(FZ_TG, a.n_graphs - x.g0);
  x.jc = min(x.lane & 15, ng - 1);
  const int r_begin = x.g0 * a.N, nrows = ng * a.N;
  if (id.m + xg.K * x.wv < a.N) fused_bwd_split_body<F, L, true, TS>(a, xg, x, nrows, r_begin);
  else fused_bwd_split_body<F, L, false, TS>(a, xg, x, nrows, r_begin);
}

}  // namespace v2x
