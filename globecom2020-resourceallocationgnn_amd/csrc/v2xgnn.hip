// Host side of libv2xgnn.so: the C ABI declared in include/v2xgnn.h.
// Owns parameters, optimizer state and activation workspaces in HBM, sequences the kernels of
// kernels.hpp for Model.predict / Model.fit (BS_brain.py:218-235), optionally as a hipGraph.
#include "../../include/v2xgnn.h"
#include "kernels.hpp"
#include "kernels_wide.hpp"
#include "kernels_fused.hpp"
#include "kernels_fused_split.hpp"
#include "kernels_mlpwg.hpp"
#include "kernels_mlpstream.hpp"
#include "kernels_small.hpp"
#include "kernels_ragged.hpp"
#include "kernels_ragged_small.hpp"
#include "host_pack.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace v2x;

namespace {

thread_local std::string g_err;

struct LayerDesc {
  int64_t off;          // offset of slot 0 in the flat parameter buffer
  int64_t slot_stride;  // floats per slot: k_real*n_out + n_out
  int k_real, n_out;    // real weight shape
  int kp, np;           // padded (multiples of 16)
  RowPad pad;           // padded K row -> real row
  int n_slabs = 0;      // partial-sum slabs the last weight-gradient launch wrote for this layer
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct ProfRec { int id; hipEvent_t ev0, ev1; };

struct GraphKey {
  int kind; const void* ptrs[12]; int sizes[6]; double scalar;
  int64_t gen;          // workspace generation of ANOTHER model whose buffers the graph bakes in (v2x_dqn_step: the target)
  bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) < 0; }
};

struct WideWgradPlan { WideWgradArgs a; int kt, nt, ntw, sp; };     // one layer's wide weight gradient: arguments + tiling

}  // namespace

struct v2x_model {
  v2x_config cfg;
  int S, Dn, De, L, F, N, C;
  std::vector<LayerDesc> gnn, dense;
  int64_t P = 0;
  float *params = nullptr, *grads = nullptr, *mom = nullptr, *vel = nullptr;
  int64_t iterations = 0;
  // activation workspace (capacity in rows)
  int64_t cap_rows = 0;
  std::vector<float*> h, a;
  float *z1 = nullptr, *z2 = nullptr, *z3 = nullptr, *q = nullptr;
  float *dq = nullptr, *dz1 = nullptr, *dz2 = nullptr, *dz3 = nullptr, *gha = nullptr, *rowloss = nullptr;
  std::vector<float*> dpre;     // one pre-activation gradient per GNN stage (read concurrently by k_wgrad)
  unsigned* nbmask = nullptr;            // [cap_rows] non-neighbour masks: fused forward -> fused backward (complement form)
  bool frag_live = false;                // h_L, a_L (and then gha) of the last training forward are fragment-major (frag_layout)
  unsigned short* gate_bits = nullptr;   // ReLU' gates of the fused forward for the fused backward: [L][gate_stride]
  int64_t gate_stride = 0;               // ushorts per stage: N x ceil(B / 16) x 64 <= 4 R + 64 N
  hipStream_t side = nullptr;   // weight-gradient kernels run here, forked/joined around the data chain
  hipStream_t cap = nullptr;    // hipGraphs are recorded on this stream and launched on the caller's (run_maybe_graph)
  std::vector<hipEvent_t> ev;   // fork / per-stage / join events of the side stream
  float* loss_dev = nullptr;
  float* loss_part = nullptr;  // 64 partial sums + the arrival counter of the split loss reduction
  float* zero_buf = nullptr;   // 4 KiB of zeros
  float* slab = nullptr; int slab_cap = 0;
  // staging for host-side inputs
  DevBuf st_xe, st_nbr, st_goff, st_rp, st_ci, st_y, st_q;
  DevBuf adj_mask;              // adjacency bit masks of the current batch (dense-graph aggregation)
  DevBuf plan_buf;              // work plan of the ragged fused forward (kernels_ragged.hpp): first graph of every workgroup
  int plan_len = 0;             // entries of the last plan (workgroups + 1)
  long long* ts_buf = nullptr;                  // V2X_FUSED_TS=1: phase time stamps of the fused forward (measurement)
  bool raw_params = false;                      // v2x_param_ptr was called: re-pack before every fused forward
  bool pk_stale = false;                        // the fragment-major copy must be rebuilt before the next fused forward
  bool compl_sums = true;                       // V2X_FUSED_COMPL (read at create): dense graphs aggregate through the complement
  bool ragged_fused = true, ragged_fused_bwd = true;   // V2X_RAGGED_FUSED / V2X_RAGGED_FUSED_BWD (read at create): kernels_ragged.hpp
  bool ragged_small_env = false; // V2X_RAGGED_SMALL (read at create): the small-tile ragged kernels (kernels_ragged_small.hpp)
  bool ragged_packed = true;    // V2X_RAGGED_PACKED: tiles packed by k_ragged_plan (0: the row-interval plan of k_adj_masks)
  bool ragged_plan_fold = true; // V2X_RAGGED_PLAN_FOLD: the packed plan as a workgroup of the mask launch when its tables are small
  bool small_predict = true;                    // V2X_SMALL_PREDICT (read at create): few-graph forwards in one launch (kernels_small.hpp)
  unsigned long long* small_h = nullptr;        // its exchange buffer [L + 1][SMALL_ROWS][F] tagged words
  unsigned long long* small_sync = nullptr;     // and per-graph departure counters [SMALL_ROWS] (64-bit)
  char* pin_h = nullptr; char* pin_d = nullptr;  // pinned, device-mapped window for host-resident few-graph predicts: the
                                                // kernel reads the batch and writes q THROUGH it (no copy launches)
  float *pk_fwd = nullptr, *pk_bwd = nullptr;   // fragment-major copies of the GNN weights (kernels_fused.hpp)
  // split-tile fused kernels (kernels_fused_split.hpp): K workgroups per 16-graph tile at the shares of the global batch
  int split_env = -1;                           // V2X_FUSED_SPLIT (read at create): -1 auto, 0 / 1 off, K forced
  unsigned long long* xchg_buf = nullptr;       // [2 L slabs][xchg_cap_tiles][N][F / 16][64][4] tagged words
  unsigned long long* xchg_sync = nullptr;      // [xchg_cap_tiles] departures
  int xchg_cap_tiles = 0;
  int* flag_host = nullptr;     // pinned, device-mapped word the kernels raise on a contract violation (tile guards,
  int* flag_dev = nullptr;      // k_validate_batch); read by the host after any synchronising call
  bool have_fwd = false;
  std::string err;
  // profiling
  bool prof = false;
  std::vector<std::string> prof_names;
  std::vector<ProfRec> prof_recs;
  // hipGraph cache
  // a captured step + the per-layer slab counts its weight-gradient launches write (host state that the launches
  // OUTSIDE the graph -- the Adam / slab-sum kernel of train_step -- depend on)
  struct GraphEntry { hipGraphExec_t exec; std::vector<int> n_slabs; bool frag_live; };
  int64_t ws_gen = 0;           // bumped whenever a workspace buffer is re-allocated (keys of graphs that bake in ANOTHER model's workspace)
  std::map<GraphKey, GraphEntry> graphs;
  bool capturing = false;
  // wide path, single-GPU training: the layers' weight gradients are collected and launched as ONE grid (wide_wgrad_flush)
  float* mlp_img = nullptr;                     // padded images of the Dense layers per slot in global memory (kernels_mlpstream.hpp)
  bool mlp_stream_now = false;                  // inside a backward pass whose MLP is k_mlp_stream: ALL Dense weight gradients are roles of k_wgrad
  bool dense0_out_now = false;                  // inside a backward pass whose MLP launch leaves Dense-0's weight gradient to k_wgrad
  // inside a DQN replay step whose MLP launch forms the targets itself (MlpArgs::tq): replaced entries, actions, where y goes
  const float* dqn_tq = nullptr; float* dqn_y = nullptr;
  bool wide_merge_now = false;                  // inside a backward pass that merges
  bool bucketed = false;                        // data parallelism wants each layer's gradient as soon as it is final
  bool fuse_adam_now = false;                   // ... and applies Adam in the weight-gradient epilogues (WideWgradArgs::adam)
  float* adam_scal = nullptr;                   // {lr_t, beta1, beta2, eps} of the current step, in device memory
  bool prof_no_fuse = false;
  std::vector<WideWgradPlan> wide_roles_buf;
  std::vector<WideWgradPlan>* wide_roles = nullptr;      // non-null: wide_wgrad collects instead of launching
};

namespace {

#define FAIL(m, code, ...)                                   \
  do {                                                       \
    char _b[512];                                            \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
    if (m) (m)->err = _b; else g_err = _b;                   \
    return code;                                             \
  } while (0)

#define HIPCHK(m, call)                                                                     \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) FAIL(m, V2X_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define CHK(x)                 \
  do {                         \
    int _r = (x);              \
    if (_r != V2X_OK) return _r; \
  } while (0)

// Captured graphs bake in the addresses of the workspace buffers: whenever one of them is re-allocated every cached
// graph is dropped (it would replay into freed memory and its outputs would no longer be where the caller reads them).
void drop_graphs(v2x_model* m) {
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second.exec);
  m->graphs.clear();
  m->ws_gen += 1;               // every caller is about to re-allocate (or has invalidated) something captured graphs point at
}

int ensure(v2x_model* m, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return V2X_OK;
  if (m->capturing) FAIL(m, V2X_ESTATE, "buffer growth during graph capture");
  drop_graphs(m);
  if (b.p) HIPCHK(m, hipFree(b.p));
  b.p = nullptr; b.cap = 0;
  HIPCHK(m, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return V2X_OK;
}

template <typename T>
int dev_alloc(v2x_model* m, T** p, size_t n) {
  HIPCHK(m, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return V2X_OK;
}

int prof_id(v2x_model* m, const char* name) {
  for (size_t i = 0; i < m->prof_names.size(); ++i)
    if (m->prof_names[i] == name) return (int)i;
  m->prof_names.push_back(name);
  return (int)m->prof_names.size() - 1;
}

// launch wrapper: optional HIP-event bracket per kernel for in-situ timing
#define LAUNCH(m, kname, kern, grid, lds, stream, args)                                  \
  do {                                                                                  \
    ProfRec _r;                                                                         \
    const bool _p = (m) && (m)->prof && !(m)->capturing;                                \
    if (_p) {                                                                           \
      _r.id = prof_id(m, kname);                                                          \
      hipEventCreate(&_r.ev0); hipEventCreate(&_r.ev1);                                     \
      hipEventRecord(_r.ev0, stream);                                                     \
    }                                                                                   \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, args);                       \
    if (_p) { hipEventRecord(_r.ev1, stream); (m)->prof_recs.push_back(_r); }             \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) FAIL(m, V2X_EHIP, "launch %s failed: %s", kname, hipGetErrorString(_e)); \
  } while (0)

// ------------------------------------------------------------------------------------ layout
void build_layout(v2x_model* m) {
  const int F = m->F, Dn = m->Dn, De = m->De, C = m->C;
  const int xr = Dn + De;   // real rows of the packed xe block
  int64_t off = 0;
  auto push = [&](std::vector<LayerDesc>& v, int k_real, int n_out, int kp, int np, RowPad pad) {
    LayerDesc d;
    d.off = off; d.k_real = k_real; d.n_out = n_out; d.kp = kp; d.np = np; d.pad = pad;
    d.slot_stride = (int64_t)k_real * n_out + n_out;
    off += d.slot_stride * m->S;
    v.push_back(d);
  };
  push(m->gnn, xr + F, F, XE + F, F, RowPad{xr, XE - xr, xr + F});
  for (int s = 1; s <= m->L; ++s)
    push(m->gnn, 2 * F + xr, F, 2 * F + XE, F, RowPad{F + xr, XE - xr, 2 * F + xr});
  push(m->dense, 2 * F + Dn, H1, 2 * F + XE, H1, RowPad{F + Dn, XE - Dn, 2 * F + Dn});
  push(m->dense, H1, H2, H1, H2P, RowPad{H1, 0, H1});
  push(m->dense, H2, H3, H2P, H3P, RowPad{H2, H2P - H2, H2});
  push(m->dense, H3, C, H3P, CP, RowPad{H3, H3P - H3, H3});
  m->P = off;
}

// dynamic LDS above 64 KiB has to be opted into per kernel; done once per model at create time so
// that nothing but kernel launches happens inside a stream capture
void allow_big_lds(const void* f) { hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }

template <int F>
void set_attrs_f() {
  allow_big_lds((const void*)k_gemm_rows<F, false, false, false>);
  allow_big_lds((const void*)k_gemm_rows<F, false, true, false>);
  allow_big_lds((const void*)k_gemm_rows<F, true, true, false>);
  allow_big_lds((const void*)k_gemm_rows<F, true, false, true>);
  allow_big_lds((const void*)k_mlp_fwd<F>);
  allow_big_lds((const void*)k_mlp_bwd<F>);
  allow_big_lds((const void*)k_mlp_train<F>);
  allow_big_lds((const void*)k_mlp_train_wg<F>);
  allow_big_lds((const void*)k_mlp_train_wg<F, true>);
  allow_big_lds((const void*)k_wgrad<F, 0>);
  allow_big_lds((const void*)k_wgrad<F, 1>);
  allow_big_lds((const void*)k_wgrad<F, 2>);
  if (F == 64) { allow_big_lds((const void*)k_wgrad<64, 4>); allow_big_lds((const void*)k_mlp_image<64>); }
}

template <int F>
void set_attrs_fused() {
  allow_big_lds((const void*)k_gnn_fwd_fused<F, 1>); allow_big_lds((const void*)k_gnn_fwd_fused<F, 1, false, true>);
  allow_big_lds((const void*)k_gnn_fwd_fused<F, 2>); allow_big_lds((const void*)k_gnn_fwd_fused<F, 2, false, true>);
  allow_big_lds((const void*)k_gnn_fwd_fused<F, 3>); allow_big_lds((const void*)k_gnn_fwd_fused<F, 3, false, true>);
  allow_big_lds((const void*)k_gnn_fwd_fused<F, 4>); allow_big_lds((const void*)k_gnn_fwd_fused<F, 4, false, true>);
  allow_big_lds((const void*)k_gnn_bwd_fused<F, 1>); allow_big_lds((const void*)k_gnn_bwd_fused<F, 1, false, true>);
  allow_big_lds((const void*)k_gnn_bwd_fused<F, 2>); allow_big_lds((const void*)k_gnn_bwd_fused<F, 2, false, true>);
  allow_big_lds((const void*)k_gnn_bwd_fused<F, 3>); allow_big_lds((const void*)k_gnn_bwd_fused<F, 3, false, true>);
  allow_big_lds((const void*)k_gnn_bwd_fused<F, 4>); allow_big_lds((const void*)k_gnn_bwd_fused<F, 4, false, true>);
  allow_big_lds((const void*)k_gnn_fwd_split<F, 1>); allow_big_lds((const void*)k_gnn_bwd_split<F, 1>);
  allow_big_lds((const void*)k_gnn_fwd_split<F, 2>); allow_big_lds((const void*)k_gnn_bwd_split<F, 2>);
  allow_big_lds((const void*)k_gnn_fwd_split<F, 3>); allow_big_lds((const void*)k_gnn_bwd_split<F, 3>);
  if (F == 64) {
    allow_big_lds((const void*)k_gnn_fwd_split<64, 2, true>); allow_big_lds((const void*)k_gnn_bwd_split<64, 2, true>);
    allow_big_lds((const void*)k_gnn_fwd_fused<64, 3, true>); allow_big_lds((const void*)k_gnn_bwd_fused<64, 3, true>);
    allow_big_lds((const void*)k_gnn_fwd_fused<64, 3, true, true>); allow_big_lds((const void*)k_gnn_bwd_fused<64, 3, true, true>);
  }
}

void set_attrs(int F) {
  if (F == 16) set_attrs_fused<16>();
  if (F == 32) set_attrs_fused<32>();
  if (F == 64) set_attrs_fused<64>();
  allow_big_lds((const void*)k_agg<false>);
  allow_big_lds((const void*)k_agg<true>);
  allow_big_lds((const void*)k_agg_dense<false>);
  allow_big_lds((const void*)k_agg_dense<true>);
  if (F == 16) set_attrs_f<16>();
  if (F == 32) set_attrs_f<32>();
  if (F == 64) set_attrs_f<64>();
  if (F >= 128) {                     // wide path: tail MLP kernels + the narrow weight-gradient kernel for Dense 1..3
    if (getenv("V2X_DEBUG_OCC")) {
      int n1 = 0, n2 = 0, n3 = 0, n4 = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k_wide_gemm<false, 4>, 256, 0);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k_wide_gemm<true, 4>, 256, 0);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n3, k_wide_wgrad_multi, 256, 0);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, k_agg_dense<false>, 256, 36 * 1024);
      hipFuncAttributes fa;
      hipFuncGetAttributes(&fa, (const void*)k_wide_gemm<false, 4>);
      fprintf(stderr, "occupancy (workgroups per CU): k_wide_gemm<false,4> %d, <true,4> %d, k_wide_wgrad_multi %d, k_agg_dense(36 KB) %d; "
              "k_wide_gemm<false,4>: %d regs, %zu B static LDS, maxDynamic %d\n", n1, n2, n3, n4, fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
    }
    allow_big_lds((const void*)k_mlp_fwd<0>);
    allow_big_lds((const void*)k_mlp_bwd<0>);
    allow_big_lds((const void*)k_wgrad<64, 1>);
  }
}

inline bool is_wide(const v2x_model* m) { return m->F >= 128; }

struct RowMapH { int n_idx, row_stride, base_mul, grid_y; };
RowMapH row_map(const v2x_model* m, int n_rows) {
  RowMapH r;
  if (m->S == 1) { r.n_idx = n_rows; r.row_stride = 1; r.base_mul = 0; r.grid_y = 1; }
  else { r.n_idx = n_rows / m->N; r.row_stride = m->N; r.base_mul = 1; r.grid_y = m->N; }
  return r;
}

int ensure_rows(v2x_model* m, int64_t R) {
  if (R <= m->cap_rows) return V2X_OK;
  if (m->capturing) FAIL(m, V2X_ESTATE, "workspace growth during graph capture");
  drop_graphs(m);
  auto re = [&](float*& p, int64_t w) -> int {
    if (p) HIPCHK(m, hipFree(p));
    p = nullptr;
    HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&p), (size_t)(R * w) * sizeof(float)));
    return V2X_OK;
  };
  const int F = m->F;
  for (int s = 0; s <= m->L; ++s) { CHK(re(m->h[s], F)); CHK(re(m->a[s], F)); }
  CHK(re(m->z1, H1)); CHK(re(m->z2, H2)); CHK(re(m->z3, H3)); CHK(re(m->q, m->C));
  CHK(re(m->dq, m->C)); CHK(re(m->dz1, H1)); CHK(re(m->dz2, H2)); CHK(re(m->dz3, H3));
  CHK(re(m->gha, 2 * F)); CHK(re(m->rowloss, 1));
  for (int s = 0; s <= m->L; ++s) CHK(re(m->dpre[s], F));
  if (m->gate_bits) HIPCHK(m, hipFree(m->gate_bits));
  m->gate_bits = nullptr;
  if (m->nbmask) HIPCHK(m, hipFree(m->nbmask));
  m->nbmask = nullptr;
  HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&m->nbmask), (size_t)R * sizeof(unsigned)));
  m->gate_stride = 4 * R + 64 * (int64_t)m->N;
  HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&m->gate_bits), (size_t)((m->L + 1) * m->gate_stride) * sizeof(unsigned short)));
  m->cap_rows = R;
  return V2X_OK;
}

// Weight-gradient decomposition.  A role (= one layer) is cut into `nc` row chunks per slot, one workgroup and
// one partial-sum slab each.  A workgroup walks its chunk sequentially (4 waves, 16-row blocks round-robin), so
// the launch is as long as its longest workgroup: the chunk count of every role fused into a launch is chosen
// proportional to the role's MFMA work (KT*NT tiles) so that the whole launch is about ONE balanced round of
// the chip (V2X_WG_ROUNDS rounds), instead of the same chunking for a 45-tile and a 2-tile layer.
int n_cus();
int role_chunks(int n_idx, int n_slots, int work, int total_work, int* chunk_out, int dflt_rows = 768) {
  static const int rounds = getenv("V2X_WG_ROUNDS") ? atoi(getenv("V2X_WG_ROUNDS")) : 0;
  static const int env_rows = getenv("V2X_WG_CHUNK") ? atoi(getenv("V2X_WG_CHUNK")) : 0;
  const int rows = env_rows > 0 ? env_rows : dflt_rows;
  int nc;
  if (rounds > 0) {                                             // work-proportional workgroup counts
    long target = ((long)n_cus() * rounds * work + total_work / 2) / total_work;
    nc = (int)(target / n_slots);
  } else {                                                      // uniform rows per workgroup (measured best: 768)
    nc = (n_idx + rows - 1) / rows;
  }
  if (nc < 1) nc = 1;
  const int max_nc = (n_idx + 4 * WG_TR - 1) / (4 * WG_TR);                         // >= one block per wave
  if (nc > max_nc) nc = max_nc;
  if (nc < 1) nc = 1;
  int chunk = (n_idx + nc - 1) / nc;
  chunk = (chunk + 4 * WG_TR - 1) / (4 * WG_TR) * (4 * WG_TR);
  nc = (n_idx + chunk - 1) / chunk;
  if (nc < 1) nc = 1;
  *chunk_out = chunk;
  return nc;
}

int ensure_slabs(v2x_model* m, int nc) {
  if (nc <= m->slab_cap) return V2X_OK;
  if (m->capturing) FAIL(m, V2X_ESTATE, "slab growth during graph capture");
  drop_graphs(m);
  if (m->slab) HIPCHK(m, hipFree(m->slab));
  m->slab = nullptr;
  HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&m->slab), (size_t)nc * m->P * sizeof(float)));
  HIPCHK(m, hipMemset(m->slab, 0, (size_t)nc * m->P * sizeof(float)));
  m->slab_cap = nc;
  return V2X_OK;
}

// ------------------------------------------------------------------------------------ device-side error flag
struct FlagWord { int* host = nullptr; int* dev = nullptr; };
int alloc_flag(FlagWord* f) {
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return V2X_EHIP;
  memset(h, 0, 64);
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { hipHostFree(h); return V2X_EHIP; }
  f->host = (int*)h; f->dev = (int*)d;
  return V2X_OK;
}
FlagWord* global_flag() {            // the per-kernel entry points that take no model handle
  static FlagWord f;
  static const bool once = [] { alloc_flag(&f); return true; }();
  (void)once;
  return &f;
}
int* flag_dev_of(v2x_model* m) { return m ? m->flag_dev : global_flag()->dev; }

// the one-launch predict's exchange buffer and departure counters back to their initial state (synchronous)
int arm_xchg(v2x_model* m, int tiles, unsigned long long floor);
int xchg_max_count(v2x_model* m, unsigned long long* out);
size_t xchg_bytes(const v2x_model* m, int tiles) {
  return (size_t)std::max(1, 2 * m->L) * tiles * m->N * (m->F / 16) * 256 * sizeof(unsigned long long);
}
int reset_exchange(v2x_model* m) {
  if (!m->small_h && !m->xchg_buf) return V2X_OK;
  HIPCHK(m, hipDeviceSynchronize());
  if (m->small_h) {
    HIPCHK(m, hipMemset(m->small_h, 0, (size_t)(m->L + 1) * 256 * m->F * sizeof(unsigned long long)));     // tag 0 = never written
    HIPCHK(m, hipMemset(m->small_sync, 0, (size_t)256 * sizeof(unsigned long long)));
  }
  if (m->xchg_buf) {
    unsigned long long seen = 0;
    CHK(xchg_max_count(m, &seen));
    CHK(arm_xchg(m, m->xchg_cap_tiles, seen + 2));          // (+ 2: a launch that died half-way may not have advanced its tiles)
  }
  return V2X_OK;
}

// to be called after the stream was synchronised: reports and clears what the kernels raised
int check_flag(v2x_model* m) {
  volatile int* p = m ? m->flag_host : global_flag()->host;
  if (!p) return V2X_OK;
  const int v = *p;
  if (!v) return V2X_OK;
  *p = 0;
  if (v & SM_ERR_TIMEOUT) {
    // k_predict_small gave up waiting for a neighbour row (kernels_small.hpp, SM_POLL_CAP): the exchange state is not the one
    // the launch expected -- an earlier launch of this model died half-way (departure counters between two multiples of N),
    // or the caller corrupted it.  Re-arm it, so that the NEXT predict works, and report this one.
    if (m) reset_exchange(m);
    FAIL(m, V2X_ESTATE, "predict: the one-launch kernel timed out waiting for a neighbour's row (exchange buffer out of step: an "
                        "earlier launch was aborted?); the exchange was reset, this call's q is invalid%s",
         (v & SM_ERR_SOURCE) ? "; the batch also holds a source id outside its graph" : "");
  }
  if (v & FZ_ERR_XCHG) {
    // a member of a split tile (kernels_fused_split.hpp) gave up waiting for its partners' rows: same remedy
    if (m) reset_exchange(m);
    FAIL(m, V2X_ESTATE, "fused graph layers: a split-tile workgroup timed out waiting for its partners' rows (exchange out of step: an "
                        "earlier launch was aborted, or the partners were not co-resident); the exchange was reset, this call's results are invalid");
  }
  if (v & 1) FAIL(m, V2X_EINVAL, "batch: a graph has more rows / edges than max_nodes / max_edges allow (LDS tile guard); results are invalid");
  FAIL(m, V2X_EINVAL, "batch violates the layout contract:%s%s%s%s", (v >> 4) & 1 ? " graph sizes vs max_nodes / graph_off;" : "",
       (v >> 4) & 2 ? " row_ptr not monotone / edge counts vs max_edges;" : "", (v >> 4) & 4 ? " source id outside its graph;" : "",
       (v >> 4) & 8 ? " sources of a row not strictly ascending (duplicate edges);" : "");
}

// ------------------------------------------------------------------------------------ batches
struct DevBatch {
  int B, R, E, max_nodes, max_edges;
  const float* xe; const float* nbr;
  const int32_t* goff; const int32_t* rp; const int32_t* ci;
};

// Host batches are checked against the layout contract before anything is copied (O(B) for the sizes that size the
// LDS tiles, O(E) for the edge list; V2X_TRUSTED_BATCHES=1 skips the O(E) part).  Device batches: v2x_validate_batch.
int validate_host_batch(v2x_model* m, const v2x_batch* b, int n_nodes) {
  const int B = b->n_graphs, R = b->n_rows;
  const int32_t *go = b->graph_off, *rp = b->row_ptr, *ci = b->col_idx;
  if (rp[0] != 0 || rp[R] != b->n_edges) FAIL(m, V2X_EINVAL, "batch: row_ptr[0] != 0 or row_ptr[n_rows] != n_edges");
  if (go && (go[0] != 0 || go[B] != R)) FAIL(m, V2X_EINVAL, "batch: graph_off[0] != 0 or graph_off[n_graphs] != n_rows");
  static const bool trusted = getenv("V2X_TRUSTED_BATCHES") != nullptr;
  for (int g = 0; g < B; ++g) {
    const int64_t r0 = go ? go[g] : (int64_t)g * n_nodes, r1 = go ? go[g + 1] : r0 + n_nodes;
    const int64_t n = r1 - r0;
    if (n < 1 || n > b->max_nodes || r0 < 0 || r1 > R)
      FAIL(m, V2X_EINVAL, "batch: graph %d has %lld rows, max_nodes is %d", g, (long long)n, b->max_nodes);
    const int64_t e0 = rp[r0], e1 = rp[r1];
    if (e1 < e0 || e1 - e0 > b->max_edges)
      FAIL(m, V2X_EINVAL, "batch: graph %d has %lld edges, max_edges is %d", g, (long long)(e1 - e0), b->max_edges);
    if (trusted) continue;
    for (int64_t q = r0; q < r1; ++q) {
      const int a0 = rp[q], a1 = rp[q + 1];
      if (a1 < a0 || a0 < e0 || a1 > e1) FAIL(m, V2X_EINVAL, "batch: row_ptr not monotone at row %lld", (long long)q);
      int prev = -1;
      for (int e = a0; e < a1; ++e) {
        const int c = ci[e];
        if (c < 0 || c >= n) FAIL(m, V2X_EINVAL, "batch: source id %d of row %lld lies outside its %lld-node graph", c, (long long)q, (long long)n);
        if (c <= prev) FAIL(m, V2X_EINVAL, "batch: sources of row %lld are not strictly ascending (duplicate edge?)", (long long)q);
        prev = c;
      }
    }
  }
  return V2X_OK;
}

int resolve_batch(v2x_model* m, const v2x_batch* b, DevBatch* d, hipStream_t st) {
  if (!b || b->n_graphs <= 0 || b->n_rows <= 0 || !b->xe || !b->row_ptr || (b->n_edges > 0 && !b->col_idx))
    FAIL(m, V2X_EINVAL, "batch: null pointer or non-positive size");
  if (!m->cfg.variable_graphs && b->n_rows != b->n_graphs * m->N)
    FAIL(m, V2X_EINVAL, "batch: n_rows (%d) != n_graphs*n_nodes (%d*%d)", b->n_rows, b->n_graphs, m->N);
  if (m->cfg.variable_graphs && !b->graph_off) FAIL(m, V2X_EINVAL, "batch: variable_graphs needs graph_off");
  if (b->max_nodes <= 0 || b->max_edges < 0) FAIL(m, V2X_EINVAL, "batch: max_nodes/max_edges not set");
  if (!m->cfg.variable_graphs && b->max_nodes != m->N)      // max_nodes selects kernels and sizes LDS tiles: one meaning only
    FAIL(m, V2X_EINVAL, "batch: max_nodes (%d) must equal n_nodes (%d) for a fixed-size model", b->max_nodes, m->N);
  d->B = b->n_graphs; d->R = b->n_rows; d->E = b->n_edges;
  d->max_nodes = b->max_nodes; d->max_edges = b->max_edges;
  if (b->on_device) {
    d->xe = b->xe; d->nbr = b->nbr_init; d->goff = b->graph_off; d->rp = b->row_ptr; d->ci = b->col_idx;
    return V2X_OK;
  }
  CHK(validate_host_batch(m, b, m->N));
  const size_t R = b->n_rows;
  CHK(ensure(m, m->st_xe, R * XE * sizeof(float)));
  HIPCHK(m, hipMemcpyAsync(m->st_xe.p, b->xe, R * XE * sizeof(float), hipMemcpyHostToDevice, st));
  d->xe = (const float*)m->st_xe.p;
  d->nbr = nullptr;
  if (b->nbr_init) {
    CHK(ensure(m, m->st_nbr, R * m->F * sizeof(float)));
    HIPCHK(m, hipMemcpyAsync(m->st_nbr.p, b->nbr_init, R * m->F * sizeof(float), hipMemcpyHostToDevice, st));
    d->nbr = (const float*)m->st_nbr.p;
  }
  d->goff = nullptr;
  if (b->graph_off) {
    CHK(ensure(m, m->st_goff, (size_t)(b->n_graphs + 1) * 4));
    HIPCHK(m, hipMemcpyAsync(m->st_goff.p, b->graph_off, (size_t)(b->n_graphs + 1) * 4, hipMemcpyHostToDevice, st));
    d->goff = (const int32_t*)m->st_goff.p;
  }
  CHK(ensure(m, m->st_rp, (R + 1) * 4));
  HIPCHK(m, hipMemcpyAsync(m->st_rp.p, b->row_ptr, (R + 1) * 4, hipMemcpyHostToDevice, st));
  d->rp = (const int32_t*)m->st_rp.p;
  CHK(ensure(m, m->st_ci, (size_t)(b->n_edges > 0 ? b->n_edges : 1) * 4));
  if (b->n_edges > 0)
    HIPCHK(m, hipMemcpyAsync(m->st_ci.p, b->col_idx, (size_t)b->n_edges * 4, hipMemcpyHostToDevice, st));
  d->ci = (const int32_t*)m->st_ci.p;
  return V2X_OK;
}

// ------------------------------------------------------------------------------------ launchers
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// A launch covers graphs [g0, g0+ng) of the batch.  Every activation / gradient buffer is indexed by the
// ABSOLUTE node row, so disjoint ranges (micro-batches) can be in flight concurrently on different streams.
struct Range { int g0, ng; };

struct IdxMap { int idx_base, n_idx, row_stride, base_mul, grid_y; };
IdxMap idx_map(const v2x_model* m, const DevBatch& d, Range r) {
  IdxMap x;
  if (m->S == 1) {            // shared weights: the GEMM row index is the node row
    x.row_stride = 1; x.base_mul = 0; x.grid_y = 1;
    if (m->cfg.variable_graphs) { x.idx_base = 0; x.n_idx = d.R; }   // ragged batches are never split
    else { x.idx_base = r.g0 * m->N; x.n_idx = r.ng * m->N; }
  } else {                    // per-node weights: slot k owns rows b*N + k, index = graph b
    x.idx_base = r.g0; x.n_idx = r.ng; x.row_stride = m->N; x.base_mul = 1; x.grid_y = m->N;
  }
  return x;
}

int n_cus() {
  static int n = 0;
  if (!n) {
    hipDeviceProp_t p;
    int dev = 0;
    hipGetDevice(&dev);
    n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

// Large dense graphs (e.g. 100 links, in-degree 98): the contraction runs on the MFMA pipe against adjacency
// bit masks (k_agg_dense) instead of one LDS gather per edge.
bool use_dense_agg(const DevBatch& d, int F) {
  static const int dense_min_nodes = env_int("V2X_AGG_DENSE_MIN_NODES", 32);
  const size_t rows_cap = (d.max_nodes + 15) / 16 * 16, mw = (d.max_nodes + 31) / 32;
  return F >= 64 && d.max_nodes >= dense_min_nodes && (int64_t)d.max_edges * 4 >= (int64_t)d.max_nodes * d.max_nodes &&
         rows_cap * AD_LDT * 4 + rows_cap * mw * 4 + 4 * 64 * 4 <= 160 * 1024;
}

AggDenseArgs agg_dense_args(const DevBatch& d, Range r, int N, int F) {
  AggDenseArgs q;
  memset(&q, 0, sizeof(q));
  q.graph_off = d.goff; q.row_ptr = d.rp; q.col_idx = d.ci;
  q.g_base = r.g0; q.n_graphs = r.ng; q.n_nodes = N; q.F = F; q.n_fg = F / 64;
  q.rows_cap = (d.max_nodes + 15) / 16 * 16;
  q.mask_words = (d.max_nodes + 31) / 32;
  return q;
}

// bytes of the packed plan's 16-bit tables when it rides on the mask launch (AggDenseArgs::plan_cap)
size_t plan16_bytes(int n_graphs, int n_wgs) { return (size_t)(n_graphs + 1) * 4 + ((size_t)2 * (n_graphs + 1) + n_wgs + 1) * 2 + 8; }   // offsets + two tables + starts
int build_adj_masks(v2x_model* m, hipStream_t st, const AggDenseArgs& q) {
  size_t lds = (size_t)2 * q.rows_cap * q.mask_words * 4 + (size_t)(q.rows_cap + 1) * 4;   // by source + by destination + row_ptr slice
  int grid = q.n_graphs;
  if (q.plan_cap > 0) lds = std::max(lds, plan16_bytes(q.n_graphs, q.plan_n));                   // workgroup 0 makes the plan first
  LAUNCH(m, "k_adj_masks", k_adj_masks, dim3(grid), lds, st, q);
  return V2X_OK;
}

int launch_agg(v2x_model* m, hipStream_t st, const DevBatch& d, Range r, int N, int F, const float* src, int src_stride,
               const float* add, int add_stride, const float* mask, float* out, int transpose) {
  if (F < 16 || F > 256 || (F & (F - 1))) FAIL(m, V2X_EINVAL, "agg: feat_dim must be a power of two in [16,256]");
  static const bool attrs_once = [] {        // the per-kernel entry points can get here without any model
    allow_big_lds((const void*)k_agg<false>); allow_big_lds((const void*)k_agg<true>);
    allow_big_lds((const void*)k_agg_small<false, false, false>); allow_big_lds((const void*)k_agg_small<true, true, true>);
    allow_big_lds((const void*)k_agg_small<true, true, false>); allow_big_lds((const void*)k_agg_small<true, false, false>);
    allow_big_lds((const void*)k_agg_dense<false>); allow_big_lds((const void*)k_agg_dense<true>);
    return true;
  }();
  (void)attrs_once;
  if (use_dense_agg(d, F)) {
    AggDenseArgs q = agg_dense_args(d, r, N, F);
    q.src = src; q.src_stride = src_stride; q.add = add; q.add_stride = add_stride; q.mask = mask; q.out = out;
    q.err = flag_dev_of(m);
    if (m) {
      q.adj = (unsigned*)m->adj_mask.p;                 // built by run_forward for this batch
      q.adjT = q.adj + (size_t)d.R * q.mask_words;
    } else {                                            // handle-less entry point: build into a scratch buffer
      static thread_local DevBuf scratch;
      const size_t need = (size_t)2 * d.R * q.mask_words * 4;
      if (need > scratch.cap) {
        if (scratch.p) hipFree(scratch.p);
        scratch.p = nullptr; scratch.cap = 0;
        HIPCHK(m, hipMalloc(&scratch.p, need));
        scratch.cap = need;
      }
      q.adj = (unsigned*)scratch.p;
      q.adjT = q.adj + (size_t)d.R * q.mask_words;
      CHK(build_adj_masks(m, st, q));
    }
    const size_t lds = (size_t)q.rows_cap * AD_LDT * 4 + (size_t)q.rows_cap * q.mask_words * 4 + 4 * 64 * 4;   // + partial column sums
    const dim3 grid((r.ng + 7) / 8 * 8 * q.n_fg);
    if (transpose) { auto k = k_agg_dense<true>; LAUNCH(m, "k_agg_bwd", k, grid, lds, st, q); }
    else { auto k = k_agg_dense<false>; LAUNCH(m, "k_agg_fwd", k, grid, lds, st, q); }
    return V2X_OK;
  }
  AggArgs a;
  a.src = src; a.src_stride = src_stride; a.add = add; a.add_stride = add_stride; a.mask = mask; a.out = out;
  a.graph_off = d.goff; a.row_ptr = d.rp; a.col_idx = d.ci;
  a.g_base = r.g0; a.g_end = r.g0 + r.ng; a.n_nodes = N; a.F = F;
  int sh = 0; while ((4 << sh) < F) ++sh;
  a.lpr_shift = sh;
  a.mask_words = (d.max_nodes + 31) / 32;
  a.transpose = transpose;
  a.err = flag_dev_of(m);
  const int nworkers = 256 >> sh;
  const size_t per_graph = (size_t)d.max_nodes * F * 4 + (size_t)(d.max_nodes + 1) * 4 + (size_t)d.max_edges * 4 +
                           (transpose ? (size_t)d.max_nodes * a.mask_words * 4 : 0);
  int gpw = 1;
  // few graphs per workgroup => many workers per graph => short per-worker row loops (the kernel is a
  // latency chain: stage -> barrier -> LDS gathers -> store); keep >= 8 workers per graph when F allows
  static const int wpg_min = env_int("V2X_AGG_WORKERS_PER_GRAPH", 8);
  while (gpw * 2 <= nworkers && nworkers / (gpw * 2) >= wpg_min && (size_t)(gpw * 2) * per_graph <= 48 * 1024 && gpw * 2 <= r.ng) gpw *= 2;
  a.gpw = gpw;
  a.rows_cap = gpw * d.max_nodes;
  a.edges_cap = gpw * d.max_edges;
  const size_t lds = (size_t)a.rows_cap * F * 4 + (size_t)(a.rows_cap + 1) * 4 + (size_t)(gpw + 1) * 4 +
                     (size_t)a.edges_cap * 4 + (transpose ? (size_t)a.rows_cap * a.mask_words * 4 : 0);
  if (lds > 160 * 1024) FAIL(m, V2X_EINVAL, "agg: graph tile (%zu B) exceeds the 160 KiB LDS", lds);
  const dim3 grid((r.ng + gpw - 1) / gpw);
  // small tiles (the 20-link graphs of the headline configuration): everything fetched before the first wait
  static const bool small_off = getenv("V2X_AGG_NO_SMALL") != nullptr;
  const int wpg = nworkers / gpw;
  const bool small = !small_off && d.ci != nullptr && (a.rows_cap << sh) <= 1024 && a.rows_cap + 1 <= 256 && a.edges_cap <= 1024 &&
                     (d.max_nodes + wpg - 1) / wpg <= 4;
  if (small) {
#define V2X_AGG_SMALL(T, A, M, NAME) { auto k = k_agg_small<T, A, M>; LAUNCH(m, NAME, k, grid, lds, st, a); return V2X_OK; }
    if (!transpose) {
      if (!add && !mask) V2X_AGG_SMALL(false, false, false, "k_agg_fwd")
    } else {
      if (add && mask) V2X_AGG_SMALL(true, true, true, "k_agg_bwd")
      if (add && !mask) V2X_AGG_SMALL(true, true, false, "k_agg_bwd")
      if (!add && !mask) V2X_AGG_SMALL(true, false, false, "k_agg_bwd")
    }
#undef V2X_AGG_SMALL
  }
  if (transpose) { auto k = k_agg<true>; LAUNCH(m, "k_agg_bwd", k, grid, lds, st, a); }
  else { auto k = k_agg<false>; LAUNCH(m, "k_agg_fwd", k, grid, lds, st, a); }
  return V2X_OK;
}

// persistent-grid sizing: ~`wgs_per_cu` workgroups per CU in total, split evenly over the slots, so that
// every SIMD ends up with the same number of 16-row tiles (fp32 MFMA is the scarce resource)
int persistent_wgs_per_slot(int n_idx, int n_slots, int wgs_per_cu) {
  const int n_tiles = (n_idx + 15) / 16;
  int w = (n_cus() * wgs_per_cu) / n_slots;
  if (w < 1) w = 1;
  const int max_useful = (n_tiles + 3) / 4;          // at least one tile per wave
  if (w > max_useful) w = max_useful;
  if (w < 1) w = 1;
  return w;
}


// ---- wide-feature path (feat_dim >= 128): LDS-tiled GEMMs of kernels_wide.hpp -------------------------------
void set_idx(WideGemmArgs& a, const IdxMap& x) {
  a.n_idx = x.n_idx; a.row_stride = x.row_stride; a.base_mul = x.base_mul; a.idx_base = x.idx_base;
}

// Output tile width in 16-column tiles.  Measured at 100 links x 1024 graphs x 256 features: whole-width (256-column)
// tiles cut the operand re-reads 4x but leave 800 workgroups at 2 waves/SIMD and are SLOWER for the node update and
// its data gradient (343 / 295 us vs 276 / 253 us with 64-column tiles at 5 waves/SIMD: L2 absorbs the re-reads);
// Dense-0's 80 columns are one 5-tile strip instead of a 64 + 16 split (154 -> 110 us).
int wide_nt(int n_out) { return n_out <= 80 ? 5 : 4; }

int launch_check(v2x_model* m, const char* kname);
int launch_wide_gemm(v2x_model* m, hipStream_t st, WideGemmArgs& a, int grid_z, bool trans, const char* name) {
  const int nt = wide_nt(a.n_out);
  // Tiling (kernels_wide.hpp, WideTiling): 128-row tiles; when the launch is more than one round of the chip's resident
  // workgroups (five per CU) and its last round would be less than three quarters full, the row blocks of that last round run
  // as two 64-row tiles each, dispatched last (V2X_WIDE_TAIL=0: whole tiles only)
  static const int tail = env_int("V2X_WIDE_TAIL", 1);
  WideTiling t;
  t.ny = (a.n_out + 16 * nt - 1) / (16 * nt);
  t.mbs = (a.n_idx + WD_TM - 1) / WD_TM;
  t.n_rb = t.mbs * grid_z;
  t.n_full_rb = t.n_rb;
  const int T = t.n_rb * t.ny, C = 5 * n_cus(), rem = T % C;
  if (tail && T > C && rem > 0 && 4 * rem < 3 * C) t.n_full_rb = t.n_rb - rem / t.ny;
  // a launch that does not fill the chip's resident slots at all (Dense-0 at configs[3]'s share: 800 tiles of 128 x 80 on 1,280
  // slots, three workgroups per CU instead of five): every row block as two half tiles (V2X_WIDE_TAIL=2 only: measured)
  if (tail == 2 && T < C && 2 * T > C) t.n_full_rb = 0;
  const dim3 grid(t.n_full_rb * t.ny + 2 * (t.n_rb - t.n_full_rb) * t.ny);
#define V2X_WIDE_GEMM(T_, NTV) { auto k = k_wide_gemm<T_, NTV>; ProfRec _r; const bool _p = m->prof && !m->capturing;                      \
    if (_p) { _r.id = prof_id(m, name); hipEventCreate(&_r.ev0); hipEventCreate(&_r.ev1); hipEventRecord(_r.ev0, st); }                     \
    hipLaunchKernelGGL(k, grid, dim3(256), 0, st, a, t);                                                                                     \
    if (_p) { hipEventRecord(_r.ev1, st); m->prof_recs.push_back(_r); }                                                                      \
    return launch_check(m, name); }
  if (!trans) {
    if (nt == 5) V2X_WIDE_GEMM(false, 5)
    V2X_WIDE_GEMM(false, 4)
  }
  V2X_WIDE_GEMM(true, 4)
#undef V2X_WIDE_GEMM
}

// out = act([seg...] W + b) of layer `ld`
int wide_fwd(v2x_model* m, hipStream_t st, const LayerDesc& ld, const IdxMap& x, const WideSeg* segs, int n_seg,
             float* out, int out_stride, int relu, const char* name) {
  WideGemmArgs a;
  memset(&a, 0, sizeof(a));
  int k = 0;
  for (int i = 0; i < n_seg; ++i) { a.seg[i] = segs[i]; k += segs[i].width; }
  a.n_seg = n_seg;
  a.W = m->params + ld.off; a.slot_stride = ld.slot_stride; a.pad = ld.pad; a.n_real = ld.n_out;
  a.k_total = k; a.n_out = ld.n_out; a.out = out; a.out_stride = out_stride; a.relu = relu; a.has_bias = 1;
  set_idx(a, x);
  return launch_wide_gemm(m, st, a, x.grid_y, false, name);
}

// gha[R][2F] = [dpre . W[h rows]^T | dpre . W[agg rows]^T];  `skip` = real weight rows between the two blocks
int wide_dgrad(v2x_model* m, hipStream_t st, const LayerDesc& ld, const IdxMap& x, const float* dpre, int d_width,
               int skip, float* gha, const char* name) {
  WideGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.seg[0] = WideSeg{dpre, d_width, d_width}; a.n_seg = 1;
  a.W = m->params + ld.off; a.slot_stride = ld.slot_stride; a.pad = ld.pad; a.n_real = ld.n_out;
  a.k_total = d_width; a.n_out = 2 * m->F; a.split = m->F; a.skip = skip;
  a.out = gha; a.out_stride = 2 * m->F;
  set_idx(a, x);
  return launch_wide_gemm(m, st, a, x.grid_y, true, name);
}

// row splits of a wide weight gradient: enough workgroups to fill the chip; one split (= the gradient is written
// in place, no slab to sum) as soon as tiles x slots alone do that
int wide_splits(int n_idx, int n_tiles, int n_slots) {
  int sp = (3 * n_cus() / 2 + n_tiles * n_slots - 1) / (n_tiles * n_slots);
  const int max_sp = (n_idx + 255) / 256;                      // at least 256 rows per split
  if (sp > max_sp) sp = max_sp;
  return sp < 1 ? 1 : sp;
}

// arguments + tiling of one layer's wide weight gradient (and the memsets of an absent input segment's weight rows)
int wide_wgrad_plan(v2x_model* m, hipStream_t st, LayerDesc& ld, const IdxMap& x, const WideSeg* segs, const int* seg_kpad,
                    int n_seg, const float* dpre, int d_stride, int zero_row0, int zero_rows, WideWgradPlan* out) {
  WideWgradArgs& a = out->a;
  memset(&a, 0, sizeof(a));
  int k_width = 0;
  for (int i = 0; i < n_seg; ++i) k_width += segs[i].width;
  // weight gradients: 128 input features x the whole output width per workgroup (324 -> 315 us for a GNN stage,
  // 180 -> 123 us for Dense-0); the 16-wide embed layer keeps 64 x 64 tiles
  const int KW = k_width <= 64 ? 64 : 128;
  int kt = 0, ns = 0;
  // the 16-wide [x | e] segment of a wide layer rides on the workgroup of K tile 0 (WideWgradArgs::xseg) instead of
  // costing a whole 128-wide K tile of its own.  Only worth it when the launch is balanced over many workgroups per CU
  // (the merged launch): on its own a stage is 500 workgroups on 256 CUs and as long as its fullest CU either way
  // (measured: 298 us unfolded, 333 us folded -- 400 workgroups, the same two per CU, one of them 12.5 % longer)
  static const int fold = env_int("V2X_WIDE_FOLD", 1);
  for (int i = 0; i < n_seg; ++i) {
    if (fold && m->wide_merge_now && KW == 128 && n_seg > 1 && segs[i].width == XE && !a.xseg.ptr) { a.xseg = segs[i]; a.xseg_kpad = seg_kpad[i]; continue; }
    a.seg[ns] = segs[i]; a.seg_kpad[ns] = seg_kpad[i]; kt += (segs[i].width + KW - 1) / KW;
    ++ns;
  }
  a.n_seg = ns;
  a.dpre = dpre; a.d_stride = d_stride; a.n_real = ld.n_out; a.pad = ld.pad;
  a.slab = m->slab; a.slab_stride = m->P; a.layer_off = ld.off; a.slot_stride = ld.slot_stride;
  a.n_idx = x.n_idx; a.row_stride = x.row_stride; a.base_mul = x.base_mul; a.idx_base = x.idx_base;
  const int ntw = KW == 64 ? 4 : (ld.n_out <= 80 ? 5 : (ld.n_out <= 128 ? 8 : 16));   // output tile = ntw x 16 columns
  const int nt = (ld.n_out + 16 * ntw - 1) / (16 * ntw);
  const int sp = wide_splits(x.n_idx, kt * nt, x.grid_y);
  if (sp > m->slab_cap) FAIL(m, V2X_ESTATE, "wide wgrad: slabs not pre-sized (%d > %d)", sp, m->slab_cap);
  a.n_split = sp;
  a.rows_per_split = ((x.n_idx + sp - 1) / sp + 31) / 32 * 32;
  ld.n_slabs = sp;
  if (sp == 1) {            // no partial sums to add: write the gradient buffer itself (saves a 2 x P-float round trip)
    a.slab = m->grads;
    ld.n_slabs = 0;
  }
  if (m->fuse_adam_now && &ld != &m->gnn[0]) {         // (the embed layer keeps k_reduce_adam: its absent-input rows have no tile)
    if (sp != 1 || zero_rows > 0) FAIL(m, V2X_ESTATE, "wide wgrad: Adam cannot ride on a split weight gradient");
    a.adam = 1; a.param = m->params; a.mom = m->mom; a.vel = m->vel; a.adam_scal = m->adam_scal;
  }
  if (zero_rows > 0) {      // weight rows of an absent (identically zero) input segment: exact zero gradient
    const int64_t count = (int64_t)zero_rows * ld.n_out;
    for (int c = 0; c < sp; ++c) {
      float* dst = a.slab + (int64_t)c * a.slab_stride + ld.off + (int64_t)zero_row0 * ld.n_out;
      const dim3 zg((unsigned)std::min<int64_t>((count / 4 + 255) / 256, 64), x.grid_y);
      hipLaunchKernelGGL(k_zero_rows, zg, dim3(256), 0, st, dst, ld.slot_stride, count);
    }
  }
  out->kt = kt; out->nt = nt; out->ntw = ntw; out->sp = sp;
  return V2X_OK;
}

int wide_wgrad(v2x_model* m, hipStream_t st, LayerDesc& ld, const IdxMap& x, const WideSeg* segs, const int* seg_kpad,
               int n_seg, const float* dpre, int d_stride, const char* name, int zero_row0 = 0, int zero_rows = 0) {
  WideWgradPlan p;
  CHK(wide_wgrad_plan(m, st, ld, x, segs, seg_kpad, n_seg, dpre, d_stride, zero_row0, zero_rows, &p));
  if (m->wide_roles) {      // collected for the merged launch (wide_wgrad_flush)
    if ((int)m->wide_roles->size() >= WWM_ROLES) FAIL(m, V2X_ESTATE, "wide wgrad: too many roles for one launch");
    m->wide_roles->push_back(p);
    return V2X_OK;
  }
  const WideWgradArgs& a = p.a;
  const dim3 grid(p.kt, p.nt, x.grid_y * p.sp);
  if (p.ntw == 4) { auto k = k_wide_wgrad<64, 4>; LAUNCH(m, name, k, grid, 0, st, a); }
  else if (p.ntw == 5) { auto k = k_wide_wgrad<128, 5>; LAUNCH(m, name, k, grid, 0, st, a); }
  else if (p.ntw == 8) { auto k = k_wide_wgrad<128, 8>; LAUNCH(m, name, k, grid, 0, st, a); }
  else { auto k = k_wide_wgrad<128, 16>; LAUNCH(m, name, k, grid, 0, st, a); }
  return V2X_OK;
}

// launch the collected roles as one grid, heaviest workgroups first (they are dispatched in block order and the launch is
// as long as its last-finishing workgroup)
int wide_wgrad_flush(v2x_model* m, hipStream_t st, const IdxMap& x) {
  std::vector<WideWgradPlan>& roles = *m->wide_roles;
  if (roles.empty()) return V2X_OK;
  std::stable_sort(roles.begin(), roles.end(), [](const WideWgradPlan& p, const WideWgradPlan& q) {
    auto w = [](const WideWgradPlan& r) { return (r.ntw == 4 ? 64 : 128) * 16 * r.ntw; };     // MFMA work of one workgroup
    return w(p) > w(q);
  });
  WideWgradMulti mu;
  memset(&mu, 0, sizeof(mu));
  int n = 0, total = 0;
  for (const WideWgradPlan& p : roles) {
    mu.w[n] = p.a; mu.kt[n] = p.kt; mu.nt[n] = p.nt;
    mu.kind[n] = p.ntw == 16 ? WWM_128x16 : (p.ntw == 8 ? WWM_128x8 : (p.ntw == 5 ? WWM_128x5 : WWM_64x4));
    mu.start[n] = total;
    total += p.kt * p.nt * x.grid_y * p.sp;
    ++n;
  }
  for (int i = n; i <= WWM_ROLES; ++i) mu.start[i] = total;
  mu.n_roles = n;
  roles.clear();
  LAUNCH(m, "k_wgrad_wide_all", k_wide_wgrad_multi, dim3(total), 0, st, mu);
  return V2X_OK;
}

void set_idx(GemmArgs& a, const IdxMap& x) {
  a.n_idx = x.n_idx; a.row_stride = x.row_stride; a.base_mul = x.base_mul; a.idx_base = x.idx_base;
}

template <int F, bool HAS0, bool HAS2, bool DGRAD>
int launch_gemm_t(v2x_model* m, hipStream_t st, GemmArgs& a, int grid_y, const char* name) {
  constexpr int FB = F / 16;
  constexpr int KB = DGRAD ? FB : ((HAS0 ? FB : 0) + 1 + (HAS2 ? FB : 0));
  constexpr int KP = DGRAD ? (2 * F + XE) : KB * 16;
  const size_t lds = (size_t)(KP * (F + 4) + F) * 4;
  static const int wgs_per_cu = env_int("V2X_GEMM_WGS_PER_CU", 2);
  const int gx = persistent_wgs_per_slot(a.n_idx, grid_y, wgs_per_cu);
  auto k = k_gemm_rows<F, HAS0, HAS2, DGRAD>;
  LAUNCH(m, name, k, dim3(gx, grid_y), lds, st, a);
  return V2X_OK;
}

template <int F>
int launch_node_fwd_f(v2x_model* m, hipStream_t st, int stage, GemmArgs& a, int grid_y) {
  if (stage == 0) {
    if (a.seg2) return launch_gemm_t<F, false, true, false>(m, st, a, grid_y, "k_node_fwd_embed");
    return launch_gemm_t<F, false, false, false>(m, st, a, grid_y, "k_node_fwd_embed");
  }
  return launch_gemm_t<F, true, true, false>(m, st, a, grid_y, "k_node_fwd");
}

// GNNLayer forward of `stage`
int launch_node_fwd(v2x_model* m, hipStream_t st, int stage, const IdxMap& x, const float* xe, const float* h_prev,
                    const float* agg_prev, float* out) {
  const LayerDesc& ld = m->gnn[stage];
  GemmArgs a;
  a.seg0 = h_prev; a.seg0_stride = m->F; a.xe = xe; a.seg2 = agg_prev; a.seg2_stride = m->F;
  a.W = m->params + ld.off; a.slot_stride = ld.slot_stride; a.pad = ld.pad;
  a.out = out; a.out_stride = m->F; a.relu = stage < m->L ? 1 : 0;
  set_idx(a, x);
  if (stage > 0 && (!h_prev || !agg_prev)) FAIL(m, V2X_EINVAL, "node_update: stage>0 needs h_prev and agg_prev");
  if (is_wide(m)) {
    const int F = m->F;
    WideSeg s[3];
    int n = 0;
    if (stage > 0) { s[n++] = WideSeg{h_prev, F, F}; s[n++] = WideSeg{xe, XE, XE}; s[n++] = WideSeg{agg_prev, F, F}; }
    else { s[n++] = WideSeg{xe, XE, XE}; if (agg_prev) s[n++] = WideSeg{agg_prev, F, F}; }
    return wide_fwd(m, st, ld, x, s, n, out, F, a.relu, stage ? "k_node_fwd" : "k_node_fwd_embed");
  }
  switch (m->F) {
    case 16: return launch_node_fwd_f<16>(m, st, stage, a, x.grid_y);
    case 32: return launch_node_fwd_f<32>(m, st, stage, a, x.grid_y);
    case 64: return launch_node_fwd_f<64>(m, st, stage, a, x.grid_y);
  }
  FAIL(m, V2X_EINVAL, "unsupported feat_dim %d", m->F);
}

// data gradient of stage >= 1: gha[R][2F] = [dpre.W1h^T | dpre.W3^T]
int launch_dgrad(v2x_model* m, hipStream_t st, int stage, const IdxMap& x, const float* dpre, float* gha) {
  const LayerDesc& ld = m->gnn[stage];
  GemmArgs a;
  a.seg0 = dpre; a.seg0_stride = m->F; a.xe = nullptr; a.seg2 = nullptr; a.seg2_stride = 0;
  a.W = m->params + ld.off; a.slot_stride = ld.slot_stride; a.pad = ld.pad;
  a.out = gha; a.out_stride = 2 * m->F; a.relu = 0;
  set_idx(a, x);
  if (is_wide(m)) return wide_dgrad(m, st, ld, x, dpre, m->F, m->Dn + m->De, gha, "k_node_dgrad");
  switch (m->F) {
    case 16: return launch_gemm_t<16, true, false, true>(m, st, a, x.grid_y, "k_node_dgrad");
    case 32: return launch_gemm_t<32, true, false, true>(m, st, a, x.grid_y, "k_node_dgrad");
    case 64: return launch_gemm_t<64, true, false, true>(m, st, a, x.grid_y, "k_node_dgrad");
  }
  FAIL(m, V2X_EINVAL, "unsupported feat_dim %d", m->F);
}

// rows per slot of the slot-major MLP internals (z2, z3, dz2, dz3, dq, rowloss): the workspace capacity, so that the
// kernel that writes them and the weight-gradient launch that reads them agree whatever the batch size
int64_t srow_stride(const v2x_model* m) { return m->S == 1 ? 0 : m->cap_rows / m->N; }

void mlp_args(v2x_model* m, MlpArgs& a, const IdxMap& x, const float* xe, const float* h, const float* agg) {
  memset(&a, 0, sizeof(a));
  a.srow_stride = srow_stride(m);
  a.h = h; a.xe = xe; a.agg = agg;
  for (int i = 0; i < 4; ++i) { a.W[i] = m->params + m->dense[i].off; a.slot_stride[i] = m->dense[i].slot_stride; }
  a.C = m->C;
  a.z1 = m->z1; a.z2 = m->z2; a.z3 = m->z3; a.q = m->q;
  a.dq = m->dq; a.dz1 = m->dz1; a.dz2 = m->dz2; a.dz3 = m->dz3; a.gha = m->gha; a.rowloss = m->rowloss;
  a.n_idx = x.n_idx; a.row_stride = x.row_stride; a.base_mul = x.base_mul; a.idx_base = x.idx_base;
}

template <int F>
int launch_mlp_f(v2x_model* m, hipStream_t st, MlpArgs& a, int grid_y, bool bwd) {
  const size_t lds = (size_t)MlpLds<F>::TOTAL * 4;
  static const int wgs_per_cu = env_int("V2X_MLP_WGS_PER_CU", 2);
  const int gx = persistent_wgs_per_slot(a.n_idx, grid_y, wgs_per_cu);
  if (!bwd) { auto k = k_mlp_fwd<F>; LAUNCH(m, "k_mlp_fwd", k, dim3(gx, grid_y), lds, st, a); }
  else { auto k = k_mlp_bwd<F>; LAUNCH(m, "k_mlp_bwd", k, dim3(gx, grid_y), lds, st, a); }
  return V2X_OK;
}

// forward + Huber + backward of the decision MLP in one launch (narrow features only)
template <int F>
int launch_mlp_train_f(v2x_model* m, hipStream_t st, MlpArgs& a, int grid_y) {
  const size_t lds = (size_t)MlpLds<F>::TOTAL * 4;
  static const int wgs_per_cu = env_int("V2X_MLP_WGS_PER_CU", 2);
  const int gx = persistent_wgs_per_slot(a.n_idx, grid_y, wgs_per_cu);
  auto k = k_mlp_train<F>;
  LAUNCH(m, "k_mlp_train", k, dim3(gx, grid_y), lds, st, a);
  return V2X_OK;
}

bool mlp_fused_training(const v2x_model* m) {
  static const bool off = getenv("V2X_MLP_SPLIT") != nullptr;
  return !off && m->F <= 64;
}

int launch_mlp_train(v2x_model* m, hipStream_t st, MlpArgs& a) {
  const int gy = m->S == 1 ? 1 : m->N;
  switch (m->F) {
    case 16: return launch_mlp_train_f<16>(m, st, a, gy);
    case 32: return launch_mlp_train_f<32>(m, st, a, gy);
    case 64: return launch_mlp_train_f<64>(m, st, a, gy);
  }
  FAIL(m, V2X_EINVAL, "unsupported feat_dim %d", m->F);
}

// ... and with the four Dense weight gradients in the same pass (kernels_mlpwg.hpp): one workgroup per CU, each
// writes one partial-sum slab of the Dense layers
bool mlp_wg_path(const v2x_model* m) {
  static const bool off = env_int("V2X_MLP_WG", 1) == 0;
  return !off && mlp_fused_training(m);
}

// work split of k_mlp_train_wg (MlpWgArgs): one workgroup per CU, equal shares of the slot-major tile list
struct MlpWgSplit { int tiles_per_slot, slot_span, tiles_per_wg, n_wgs, n_slabs; };
MlpWgSplit mlp_wg_split(int n_idx, int n_slots) {
  MlpWgSplit s;
  const int T = (n_idx + 15) / 16, cus = n_cus();
  s.tiles_per_slot = T;
  auto up4 = [](int v) { return (v + 3) / 4 * 4; };          // whole rounds of the 4 waves
  // whole workgroups per slot (see MlpWgArgs for the packed alternative that was tried)
  int gx = std::max(1, std::min(cus / std::max(n_slots, 1), (T + 3) / 4));
  const int qa = up4((T + gx - 1) / gx);
  gx = (T + qa - 1) / qa;
  s.slot_span = gx * qa; s.tiles_per_wg = qa;
  s.n_wgs = gx * n_slots;
  s.n_slabs = gx;
  return s;
}

template <int F>
int launch_mlp_train_wg_f(v2x_model* m, hipStream_t st, const MlpArgs& a, int n_slots, bool wg0) {
  const size_t lds = (size_t)MlpWgLds<F>::TOTAL * 4;
  const MlpWgSplit sp = mlp_wg_split(a.n_idx, n_slots);
  if (sp.n_slabs > m->slab_cap) FAIL(m, V2X_ESTATE, "mlp_train_wg: slabs not pre-sized (%d > %d)", sp.n_slabs, m->slab_cap);
  MlpTrainWgArgs t;
  memset(&t, 0, sizeof(t));
  t.a = a;
  t.w.slab = m->slab; t.w.slab_stride = m->P;
  t.w.tiles_per_slot = sp.tiles_per_slot; t.w.slot_span = sp.slot_span; t.w.tiles_per_wg = sp.tiles_per_wg; t.w.n_slots = n_slots; t.w.n_slabs = sp.n_slabs;
  t.w.ts = m->ts_buf ? m->ts_buf + 2 * 8 * 64 : nullptr;
  for (int i = 0; i < 4; ++i) {
    LayerDesc& ld = m->dense[i];
    ld.n_slabs = sp.n_slabs;              // remembered for the slab reduction
    t.w.l[i] = MlpWgLayer{ld.off, ld.slot_stride, ld.n_out, ld.pad};
  }
  if constexpr (F == 64) {
    if (!wg0) {             // Dense-0's weight gradient is a role of the graph layers' launch (dense0_rides): dz1 rows instead
      if (a.frag_groups > 0) { auto k = k_mlp_train_wg<F, true, false>; LAUNCH(m, "k_mlp_train_wg123", k, dim3(sp.n_wgs), lds, st, t); }
      else { auto k = k_mlp_train_wg<F, false, false>; LAUNCH(m, "k_mlp_train_wg123", k, dim3(sp.n_wgs), lds, st, t); }
      return V2X_OK;
    }
  }
  if (!wg0) FAIL(m, V2X_ESTATE, "mlp_train_wg: Dense-0 outside the launch needs feat_dim 64");
  if (a.frag_groups > 0) { auto k = k_mlp_train_wg<F, true>; LAUNCH(m, "k_mlp_train_wg", k, dim3(sp.n_wgs), lds, st, t); }
  else { auto k = k_mlp_train_wg<F>; LAUNCH(m, "k_mlp_train_wg", k, dim3(sp.n_wgs), lds, st, t); }
  return V2X_OK;
}

int launch_mlp_train_wg(v2x_model* m, hipStream_t st, const MlpArgs& a, bool wg0 = true) {
  const int gy = m->S == 1 ? 1 : m->N;
  switch (m->F) {
    case 16: return launch_mlp_train_wg_f<16>(m, st, a, gy, wg0);
    case 32: return launch_mlp_train_wg_f<32>(m, st, a, gy, wg0);
    case 64: return launch_mlp_train_wg_f<64>(m, st, a, gy, wg0);
  }
  FAIL(m, V2X_EINVAL, "unsupported feat_dim %d", m->F);
}

int launch_mlp(v2x_model* m, hipStream_t st, MlpArgs& a, bool bwd) {
  const int gy = m->S == 1 ? 1 : m->N;
  if (is_wide(m)) {
    // Dense-0 and its data gradient are wide GEMMs; Dense 1..3 (+ Huber) stay register-chained ("tail" form)
    IdxMap x;
    x.idx_base = a.idx_base; x.n_idx = a.n_idx; x.row_stride = a.row_stride; x.base_mul = a.base_mul; x.grid_y = gy;
    const int F = m->F;
    if (!bwd) {
      WideSeg s[3] = {WideSeg{a.h, F, F}, WideSeg{a.xe, XE, XE}, WideSeg{a.agg, F, F}};
      CHK(wide_fwd(m, st, m->dense[0], x, s, 3, m->z1, H1, 1, "k_dense0_fwd"));
      return launch_mlp_f<0>(m, st, a, gy, false);
    }
    CHK(launch_mlp_f<0>(m, st, a, gy, true));
    return wide_dgrad(m, st, m->dense[0], x, m->dz1, H1, m->Dn, m->gha, "k_dense0_dgrad");
  }
  switch (m->F) {
    case 16: return launch_mlp_f<16>(m, st, a, gy, bwd);
    case 32: return launch_mlp_f<32>(m, st, a, gy, bwd);
    case 64: return launch_mlp_f<64>(m, st, a, gy, bwd);
  }
  FAIL(m, V2X_EINVAL, "unsupported feat_dim %d", m->F);
}

// weight gradients: build the role description of one layer ...
int layer_work(const LayerDesc& ld) { return (ld.kp / 16) * (ld.np / 16); }

int wgrad_role(v2x_model* m, LayerDesc& ld, int kind, const IdxMap& x, int total_work, const WgSeg* segs, int n_seg,
               const float* dpre, int d_stride, WgradArgs& a, int rows_override = 0) {
  int chunk;
  const bool gnn_kind = kind < WG_KIND_DENSE0 || kind >= WG_KIND_EMBED_NONBR;
  // measured at batch 4096 x 20 nodes in the merged launch: the SAME 1024 rows per workgroup for every role (87 us) beats
  // 1024 / 768 for the GNN / Dense families (102 us, the optimum when the two families were separate launches)
  // (the graph layers on their own -- the Dense gradients come out of k_mlp_train_wg -- : 896, i.e. 5 x 832 rows per
  //  slot at batch 4096: 46.8 us against 53.1 at 1024, 48.7 at 768, 49.4 at 704, 57.2 at 640)
  static const int rows_gnn = env_int("V2X_WG_CHUNK_GNN", 0), rows_dense = env_int("V2X_WG_CHUNK_DENSE", 1024);
  int rows_g = rows_gnn > 0 ? rows_gnn : (mlp_wg_path(m) ? 896 : 1024);
  static const int rows_embed = env_int("V2X_WG_CHUNK_EMBED", 0);       // the (light) embed role on its own chunking
  if (rows_embed > 0 && (kind == WG_KIND_EMBED || kind == WG_KIND_EMBED_NONBR)) rows_g = rows_embed;
  if (rows_override > 0) rows_g = rows_override;
  const int nc = role_chunks(x.n_idx, x.grid_y, layer_work(ld), total_work, &chunk, (gnn_kind || rows_override > 0) ? rows_g : rows_dense);
  if (nc > m->slab_cap) FAIL(m, V2X_ESTATE, "wgrad: slabs not pre-sized (%d > %d)", nc, m->slab_cap);
  ld.n_slabs = nc;                       // remembered for the slab reduction
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n_seg; ++i) a.seg[i] = segs[i];
  a.n_seg = n_seg;
  a.dpre = dpre; a.d_stride = d_stride; a.n_real = ld.n_out;
  a.kp = ld.kp; a.np = ld.np; a.pad = ld.pad;
  a.slab = m->slab; a.slab_stride = m->P; a.layer_off = ld.off; a.slot_stride = ld.slot_stride;
  a.n_idx = x.n_idx; a.row_stride = x.row_stride; a.base_mul = x.base_mul; a.chunk = chunk;
  a.idx_base = x.idx_base; a.chunk_base = 0; a.n_chunks = nc; a.kind = kind;
  a.dpre_slot_major = (kind >= WG_KIND_DENSE1 && kind <= WG_KIND_DENSE3) ? 1 : 0;      // dz2, dz3, dq (MlpArgs::srow_stride)
  a.srow_stride = (int)srow_stride(m);
  a.zeros = m->zero_buf;
  a.ts = (m->ts_buf && (kind == WG_KIND_GNN || kind >= WG_KIND_GNN_E1)) ? m->ts_buf + 3 * 8 * 64 : nullptr;
  return V2X_OK;
}

// ... and launch up to WG_MAX_ROLES of them as one grid (blockIdx.z = role)
int launch_wgrad_multi(v2x_model* m, hipStream_t st, const IdxMap& x, WgradMulti& mu, int n_roles, const char* name) {
  int maxt = 1, nc = 1;
  for (int i = 0; i < n_roles; ++i) {
    maxt = std::max(maxt, (mu.w[i].kp / 16) * (mu.w[i].np / 16) + (mu.w[i].kind >= WG_KIND_GNN_E1 ? 4 : 0));
    nc = std::max(nc, mu.w[i].n_chunks);
  }
  {                                     // phase stamps (V2X_FUSED_TS=1): of ONE role, V2X_WG_TS_ROLE (default: the first)
    const int ts_role = env_int("V2X_WG_TS_ROLE", 0);
    for (int i = 0; i < n_roles; ++i)
      if (i != ts_role) mu.w[i].ts = nullptr;
  }
  const size_t lds = (size_t)(2 * maxt * 64 * 4 + 4 * 9 * 16) * 4;      // accumulator exchange (two sets) + bias (per wave)
  dim3 grid(nc, x.grid_y, n_roles);
  {                                     // unequal chunk counts: a packed 1-D grid of the real workgroups (kernels.hpp WgradMulti)
    bool uneven = false;
    for (int i = 0; i < n_roles; ++i) uneven = uneven || mu.w[i].n_chunks != nc || mu.w[i].chain > 0;
    mu.packed = uneven ? 1 : 0;
    int at = 0, follow = 0;                 // (the `chain` roles behind a role are run by ITS workgroups: none of their own)
    for (int i = 0; i <= WG_MAX_ROLES; ++i) {
      mu.wg_begin[i] = at;
      if (i < n_roles) {
        if (follow > 0) --follow;
        else { at += mu.w[i].n_chunks * x.grid_y; follow = mu.w[i].chain; }
      }
    }
    if (uneven) grid = dim3(at, 1, 1);
  }
  bool dense = false, gnn = false;
  bool halves = false;
  for (int i = 0; i < n_roles; ++i) {
    if (mu.w[i].kind >= WG_KIND_DENSE0A) { halves = true; continue; }
    ((mu.w[i].kind >= WG_KIND_DENSE0 && mu.w[i].kind <= WG_KIND_DENSE3) ? dense : gnn) = true;
  }
#define V2X_WG_CASE(FF)                                                              \
  if (m->F == FF) {                                                                  \
    if (halves && dense) {                                                           \
      if constexpr (FF == 64) { auto k = k_wgrad<64, 4>; LAUNCH(m, name, k, grid, lds, st, mu); }  \
      else FAIL(m, V2X_ESTATE, "wgrad: Dense roles next to the Dense-0 halves need feat_dim 64");  \
    }                                                                                \
    else if (halves) { auto k = k_wgrad<FF, 3>; LAUNCH(m, name, k, grid, lds, st, mu); }    \
    else if (dense && gnn) { auto k = k_wgrad<FF, 2>; LAUNCH(m, name, k, grid, lds, st, mu); }  \
    else if (dense) { auto k = k_wgrad<FF, 1>; LAUNCH(m, name, k, grid, lds, st, mu); }  \
    else { auto k = k_wgrad<FF, 0>; LAUNCH(m, name, k, grid, lds, st, mu); }           \
    return V2X_OK;                                                                   \
  }
  V2X_WG_CASE(16) V2X_WG_CASE(32) V2X_WG_CASE(64)
#undef V2X_WG_CASE
  if (is_wide(m) && dense) {            // Dense 1..3 only (their operand widths do not depend on F)
    auto k = k_wgrad<64, 1>;
    LAUNCH(m, name, k, grid, lds, st, mu);
    return V2X_OK;
  }
  FAIL(m, V2X_EINVAL, "wgrad: %d output tiles unsupported", maxt);
}

int wgrad_gnn_role(v2x_model* m, int stage, const IdxMap& x, int total_work, const float* xe, const float* h_prev,
                   const float* agg_prev, const float* dpre, WgradArgs& a, int embed_tiles = 0, int rows_override = 0) {
  const int F = m->F;
  WgSeg s[3];
  int n = 0;
  if (stage > 0) { s[n++] = WgSeg{h_prev, F, F, 0, 0}; s[n++] = WgSeg{xe, XE, XE, F, 0}; s[n++] = WgSeg{agg_prev, F, F, F + XE, 0}; }
  else { s[n++] = WgSeg{xe, XE, XE, 0, 0}; s[n++] = WgSeg{agg_prev /* neighbour-init or null */, F, F, XE, 0}; }
  int kind = stage ? WG_KIND_GNN : (agg_prev ? WG_KIND_EMBED : WG_KIND_EMBED_NONBR);
  if (stage && embed_tiles) kind = embed_tiles == 1 ? WG_KIND_GNN_E1 : (embed_tiles == 2 ? WG_KIND_GNN_E2 : WG_KIND_GNN_E4);
  CHK(wgrad_role(m, m->gnn[stage], kind, x, total_work, s, n, dpre, F, a, rows_override));
  if (stage && embed_tiles) {            // + the embed layer's columns [(stage - 1) * 16 * embed_tiles, ...): kernels.hpp, wgrad_body EN
    const LayerDesc& e = m->gnn[0];
    a.e_dpre = m->dpre[0]; a.e_stride = F; a.e_col0 = (stage - 1) * 16 * embed_tiles; a.e_n_real = e.n_out;
    a.e_layer_off = e.off; a.e_slot_stride = e.slot_stride; a.e_pad = e.pad;
  }
  return V2X_OK;
}

int wide_wgrad_gnn(v2x_model* m, hipStream_t st, int stage, const IdxMap& x, const float* xe, const float* h_prev,
                   const float* agg_prev, const float* dpre) {
  const int F = m->F;
  WideSeg s[3];
  int kp[3], n = 0;
  if (stage > 0) {
    s[n] = WideSeg{h_prev, F, F}; kp[n++] = 0;
    s[n] = WideSeg{xe, XE, XE}; kp[n++] = F;
    s[n] = WideSeg{agg_prev, F, F}; kp[n++] = F + XE;
  } else {
    s[n] = WideSeg{xe, XE, XE}; kp[n++] = 0;
    if (agg_prev) { s[n] = WideSeg{agg_prev, F, F}; kp[n++] = XE; }
    // neighbour-init absent (the reference always feeds zeros): its F weight rows (real rows Dn+De ..) are memset
    else return wide_wgrad(m, st, m->gnn[0], x, s, kp, n, dpre, F, "k_wgrad_embed", m->Dn + m->De, F);
  }
  return wide_wgrad(m, st, m->gnn[stage], x, s, kp, n, dpre, F, stage ? "k_wgrad_gnn" : "k_wgrad_embed");
}

// rows per workgroup of the graph layers' roles when the embed gradient rides on them: one workgroup per CU if that
// leaves >= 64 rows each (one 16-row block per wave), else the default chunking (0).  (Until round 3 the bound was 256
// rows: a 512- or 1024-graph share of a fixed global batch -- 8 / 4 GPUs -- then fell back to 896-row chunks, i.e. 40 / 80
// workgroups walking 8 blocks per wave: 33.8 / 34.1 us per launch against 19.2 / 20.6 us with 128- / 192-row chunks.)
// extra_roles: roles of about a graph layer's weight that ride in the same grid (Dense-0 at small batches, dense0_rides)
int merged_wg_rows(const v2x_model* m, int n_idx, int n_slots, int extra_roles = 0) {
  static const int rows_env = env_int("V2X_WG_CHUNK_MERGED", 0);
  if (rows_env > 0) return rows_env;
  const int nc_fit = n_cus() / std::max(1, (m->L + extra_roles) * n_slots);
  static const int min_rows = env_int("V2X_WG_MERGED_MIN_ROWS", 64);
  if (nc_fit >= 1 && n_idx / nc_fit >= min_rows) return ((n_idx + nc_fit - 1) / nc_fit + 63) / 64 * 64;
  return 0;
}

// one GNN stage on its own (per-kernel entry point)
int wgrad_gnn(v2x_model* m, hipStream_t st, int stage, const IdxMap& x, const float* xe, const float* h_prev,
              const float* agg_prev, const float* dpre) {
  if (is_wide(m)) return wide_wgrad_gnn(m, st, stage, x, xe, h_prev, agg_prev, dpre);
  WgradMulti mu;
  memset(&mu, 0, sizeof(mu));
  CHK(wgrad_gnn_role(m, stage, x, layer_work(m->gnn[stage]), xe, h_prev, agg_prev, dpre, mu.w[0]));
  return launch_wgrad_multi(m, st, x, mu, 1, stage ? "k_wgrad_gnn" : "k_wgrad_embed");
}

// The embed layer's gradient rides on the graph layers' roles (wgrad_gnn_all)
bool embed_rides(const v2x_model* m, const DevBatch& d) {
  static const int merge_env = env_int("V2X_WG_EMBED_MERGE", 1);
  const int NTf = m->F / 16;
  return merge_env && !is_wide(m) && !d.nbr && m->L >= 1 && m->L <= NTf && NTf % m->L == 0 && m->L + 3 <= WG_MAX_ROLES;
}
// Small batches (the shares of the metric's global batch on 4 / 8 GPUs): k_mlp_train_wg is ONE wave's latency chain there (<= 2
// tiles per wave) and the graph layers' weight-gradient launch leaves a third of the chip idle: Dense-0's weight gradient -- 180
// of a tile's 796 MFMAs, 45 of the 68 accumulator tiles the MLP launch exchanges and writes at its end -- moves over as a role
// of that launch (kernels_mlpwg.hpp WG0 = false, kernels.hpp WG_KIND_DENSE0_FRAG).  V2X_MLP_WG0=1: never, =0: whenever possible.
bool dense0_rides(const v2x_model* m, const DevBatch& d, const IdxMap& x) {
  const int env = env_int("V2X_MLP_WG0", -1);                     // (read per call: the tests switch it inside one process)
  if (env == 1 || m->F != 64 || m->cfg.variable_graphs || !embed_rides(m, d)) return false;
  if (x.n_idx % 16 || x.idx_base % 16) return false;             // whole 16-row groups (the fragment-major reader)
  if (env == 0) return true;
  // tiles per workgroup = 4 x tiles per wave.  Measured (profiles/r06_dense0_role_shares.txt): 512 / 1024 / 2048 graphs of 20 links
  // (4 / 8 / 12 tiles per workgroup) 0.1201 -> 0.1131, 0.1512 -> 0.1463, 0.1929 -> 0.1880 ms per step; at 4096 (20-24 tiles) the chip
  // is full either way and the heavier weight-gradient launch costs more than the MLP launch saves (0.2554 -> 0.2658)
  const int max_tiles = env_int("V2X_MLP_WG0_TILES", 12);
  return mlp_wg_split(x.n_idx, x.grid_y).tiles_per_wg <= max_tiles;
}

// all GNN stages (needs dpre[0..L]) in ceil((L+1)/4) launches
int wgrad_gnn_all(v2x_model* m, hipStream_t st, const IdxMap& x, const DevBatch& d) {
  if (is_wide(m)) {
    if (m->wide_merge_now) m->wide_roles = &m->wide_roles_buf;          // collect: L + 1 graph layers (+ Dense-0), one grid
    int rc = V2X_OK;
    for (int s = m->L; s >= 0 && rc == V2X_OK; --s)
      rc = wide_wgrad_gnn(m, st, s, x, d.xe, s ? m->h[s - 1] : nullptr, s ? m->a[s - 1] : d.nbr, m->dpre[s]);
    if (m->wide_merge_now && rc == V2X_OK) {
      const int F = m->F, L = m->L;
      WideSeg w0[3] = {WideSeg{m->h[L], F, F}, WideSeg{d.xe, XE, XE}, WideSeg{m->a[L], F, F}};
      const int kp[3] = {0, F, F + XE};
      rc = wide_wgrad(m, st, m->dense[0], x, w0, kp, 3, m->dz1, H1, "k_wgrad_dense0");
      if (rc == V2X_OK) rc = wide_wgrad_flush(m, st, x);
    }
    m->wide_roles = nullptr;
    m->wide_roles_buf.clear();
    return rc;
  }
  WgradMulti mu;
  // The embed layer's gradient (16 MFMAs per block, load-bound) rides on the graph layers' roles when its F / 16 output
  // tiles divide evenly over the L stages: no workgroups of its own, and the heavy roles get one workgroup per CU --
  // n_cus / (L * slots) chunks per slot (batch 4096 x 20 slots x 2 stages: 6 chunks = 240 workgroups x 11 blocks per wave
  // instead of 5 chunks = 200 x 13 with 100 embed workgroups queueing behind them).
  const int NTf = m->F / 16;
  if (embed_rides(m, d)) {
    const int en = NTf / m->L;
    const bool d0 = m->dense0_out_now;                           // + Dense-0 (k_mlp_train_wg left dz1 instead of dW0)
    const int rows = merged_wg_rows(m, x.n_idx, x.grid_y, d0 ? 1 : 0);
    memset(&mu, 0, sizeof(mu));
    int total = 0, n = 0;
    for (int t = m->L; t >= 1; --t) total += layer_work(m->gnn[t]);
    for (int s = m->L; s >= 1; --s)
      CHK(wgrad_gnn_role(m, s, x, total, d.xe, m->h[s - 1], m->a[s - 1], m->dpre[s], mu.w[n++], en, rows));
    if (d0) {
      // Dense-0 in two halves along K, [h | x] (25 accumulator tiles) and [agg] (20): lighter than a graph layer's role (36 + the
      // embed's; those go first: workgroups are dispatched in role order), so they take HALF as many row chunks -- at the 512- and 1024-graph shares 4 chunks per graph-layer role and 2
      // per half is 240 workgroups, one per CU
      const int F = m->F, L = m->L;
      const int groups = m->frag_live ? d.B / FZ_TG : 0;
      int chunk_g;
      const int nc_g = role_chunks(x.n_idx, x.grid_y, 1, 1, &chunk_g, rows > 0 ? rows : 896);
      const int rows_d = (((x.n_idx + std::max(1, nc_g / 2) - 1) / std::max(1, nc_g / 2)) + 63) / 64 * 64;
      WgSeg sa[2] = {WgSeg{m->h[L], F, F, 0, 0}, WgSeg{d.xe, XE, XE, F, 0}};
      CHK(wgrad_role(m, m->dense[0], groups ? WG_KIND_DENSE0A_F : WG_KIND_DENSE0A, x, total, sa, 2, m->dz1, H1, mu.w[n], rows_d));
      mu.w[n].frag_groups = groups; mu.w[n].kp = F + XE;
      const int first_half = n++;
      // (k_mlp_stream: Dense 1..3 from the rows it left -- z1..z3, dz2, dz3, dq -- as roles WITHOUT workgroups of their own, run by
      //  the halves' workgroups behind their own bodies, WgradArgs::chain: as roles of their own in 512-row chunks they made the
      //  launch 35 us instead of 21.6 at the 512-graph share)
      if (m->mlp_stream_now) {
        WgSeg s2[1] = {WgSeg{m->z2, H2, H2, 0, 1}};
        CHK(wgrad_role(m, m->dense[2], WG_KIND_DENSE2, x, total, s2, 1, m->dz3, H3, mu.w[n++], rows_d));
        WgSeg s3[1] = {WgSeg{m->z3, H3, H3, 0, 1}};
        CHK(wgrad_role(m, m->dense[3], WG_KIND_DENSE3, x, total, s3, 1, m->dq, m->C, mu.w[n++], rows_d));
        mu.w[first_half].chain = 2;
      }
      WgSeg sb[1] = {WgSeg{m->a[L], F, F, 0, 0}};
      CHK(wgrad_role(m, m->dense[0], groups ? WG_KIND_DENSE0B_F : WG_KIND_DENSE0B, x, total, sb, 1, m->dz1, H1, mu.w[n], rows_d));
      mu.w[n].frag_groups = groups; mu.w[n].kp = F; mu.w[n].k_off = F + XE; mu.w[n].no_bias = 1;
      const int second_half = n++;
      if (m->mlp_stream_now) {
        WgSeg s1[1] = {WgSeg{m->z1, H1, H1, 0, 0}};
        CHK(wgrad_role(m, m->dense[1], WG_KIND_DENSE1, x, total, s1, 1, m->dz2, H2, mu.w[n++], rows_d));
        mu.w[second_half].chain = 1;
      }
    }
    m->gnn[0].n_slabs = m->gnn[1].n_slabs;                     // every stage role writes its columns of every embed slab
    return launch_wgrad_multi(m, st, x, mu, n, m->mlp_stream_now ? "k_wgrad_gnn_d0123" : (d0 ? "k_wgrad_gnn_d0" : "k_wgrad_gnn"));
  }
  if (m->dense0_out_now) FAIL(m, V2X_ESTATE, "wgrad: Dense-0 was left to a launch that cannot take it");
  int n = 0, s_first = m->L;
  for (int s = m->L; s >= 0; --s) {
    if (n == 0) { memset(&mu, 0, sizeof(mu)); s_first = s; }
    int total = 0;                                    // work of the stages sharing this launch
    for (int t = s_first; t >= 0 && t > s_first - WG_MAX_ROLES; --t) total += layer_work(m->gnn[t]);
    CHK(wgrad_gnn_role(m, s, x, total, d.xe, s ? m->h[s - 1] : nullptr, s ? m->a[s - 1] : d.nbr, m->dpre[s], mu.w[n]));
    if (++n == WG_MAX_ROLES || s == 0) { CHK(launch_wgrad_multi(m, st, x, mu, n, "k_wgrad_gnn")); n = 0; }
  }
  return V2X_OK;
}

// the 4 Dense layers in one launch
int wgrad_mlp(v2x_model* m, hipStream_t st, const IdxMap& x, const float* xe, const float* h, const float* agg) {
  const int F = m->F;
  int total = 0;
  for (int i = 0; i < 4; ++i) total += layer_work(m->dense[i]);
  WgradMulti mu;
  memset(&mu, 0, sizeof(mu));
  if (is_wide(m)) {
    WideSeg w0[3] = {WideSeg{h, F, F}, WideSeg{xe, XE, XE}, WideSeg{agg, F, F}};
    const int kp[3] = {0, F, F + XE};
    if (!m->wide_merge_now) CHK(wide_wgrad(m, st, m->dense[0], x, w0, kp, 3, m->dz1, H1, "k_wgrad_dense0"));   // else: a role of the merged launch
    WgSeg t1[1] = {WgSeg{m->z1, H1, H1, 0, 0}};
    CHK(wgrad_role(m, m->dense[1], WG_KIND_DENSE1, x, total, t1, 1, m->dz2, H2, mu.w[0]));
    WgSeg t2[1] = {WgSeg{m->z2, H2, H2, 0, 1}};
    CHK(wgrad_role(m, m->dense[2], WG_KIND_DENSE2, x, total, t2, 1, m->dz3, H3, mu.w[1]));
    WgSeg t3[1] = {WgSeg{m->z3, H3, H3, 0, 1}};
    CHK(wgrad_role(m, m->dense[3], WG_KIND_DENSE3, x, total, t3, 1, m->dq, m->C, mu.w[2]));
    return launch_wgrad_multi(m, st, x, mu, 3, "k_wgrad_dense");
  }
  WgSeg s0[3] = {WgSeg{h, F, F, 0, 0}, WgSeg{xe, XE, XE, F, 0}, WgSeg{agg, F, F, F + XE, 0}};
  CHK(wgrad_role(m, m->dense[0], WG_KIND_DENSE0, x, total, s0, 3, m->dz1, H1, mu.w[0]));
  WgSeg s1[1] = {WgSeg{m->z1, H1, H1, 0, 0}};
  CHK(wgrad_role(m, m->dense[1], WG_KIND_DENSE1, x, total, s1, 1, m->dz2, H2, mu.w[1]));
  WgSeg s2[1] = {WgSeg{m->z2, H2, H2, 0, 1}};
  CHK(wgrad_role(m, m->dense[2], WG_KIND_DENSE2, x, total, s2, 1, m->dz3, H3, mu.w[2]));
  WgSeg s3[1] = {WgSeg{m->z3, H3, H3, 0, 1}};
  CHK(wgrad_role(m, m->dense[3], WG_KIND_DENSE3, x, total, s3, 1, m->dq, m->C, mu.w[3]));
  return launch_wgrad_multi(m, st, x, mu, 4, "k_wgrad_dense");
}

// every layer's weight gradient in ONE launch (single-stream backward): roles heaviest first, because workgroups
// are dispatched in blockIdx.z order and the launch is as long as its last-finishing workgroup
bool wgrad_all_fits(const v2x_model* m) { return !is_wide(m) && m->L + 1 + 4 <= WG_MAX_ROLES; }

int wgrad_all(v2x_model* m, hipStream_t st, const IdxMap& x, const DevBatch& d) {
  const int F = m->F, L = m->L;
  int total = 0;
  for (int i = 0; i < 4; ++i) total += layer_work(m->dense[i]);
  for (int s = 0; s <= L; ++s) total += layer_work(m->gnn[s]);
  WgradMulti mu;
  memset(&mu, 0, sizeof(mu));
  int n = 0;
  WgSeg s0[3] = {WgSeg{m->h[L], F, F, 0, 0}, WgSeg{d.xe, XE, XE, F, 0}, WgSeg{m->a[L], F, F, F + XE, 0}};
  CHK(wgrad_role(m, m->dense[0], WG_KIND_DENSE0, x, total, s0, 3, m->dz1, H1, mu.w[n++]));
  for (int s = L; s >= 1; --s)
    CHK(wgrad_gnn_role(m, s, x, total, d.xe, m->h[s - 1], m->a[s - 1], m->dpre[s], mu.w[n++]));
  CHK(wgrad_gnn_role(m, 0, x, total, d.xe, nullptr, d.nbr, m->dpre[0], mu.w[n++]));
  WgSeg s1[1] = {WgSeg{m->z1, H1, H1, 0, 0}};
  CHK(wgrad_role(m, m->dense[1], WG_KIND_DENSE1, x, total, s1, 1, m->dz2, H2, mu.w[n++]));
  WgSeg s2[1] = {WgSeg{m->z2, H2, H2, 0, 1}};
  CHK(wgrad_role(m, m->dense[2], WG_KIND_DENSE2, x, total, s2, 1, m->dz3, H3, mu.w[n++]));
  WgSeg s3[1] = {WgSeg{m->z3, H3, H3, 0, 1}};
  CHK(wgrad_role(m, m->dense[3], WG_KIND_DENSE3, x, total, s3, 1, m->dq, m->C, mu.w[n++]));
  return launch_wgrad_multi(m, st, x, mu, n, "k_wgrad_all");
}

struct LossJob { int n_out, n_idx, stride; float scale; int64_t slot_stride; };   // n_out == 0: no loss role;
                                                                                 // rowloss(slot, i) = slot*slot_stride + i*stride

// bucket: -1 = the whole flat buffer, 0 = the Dense layers (its tail), 1 = the graph layers (its head)
int64_t bucket_begin(const v2x_model* m, int bucket) { return bucket == 0 ? m->dense[0].off : 0; }
int64_t bucket_end(const v2x_model* m, int bucket) { return bucket == 1 ? m->dense[0].off : m->P; }

// Gradient buckets of the phased forward+backward (data parallelism, v2x_forward_backward_phase), in the order in which they
// become final.  Narrow features (graph-major fused kernels: the graph layers finish together): [Dense layers], [graph
// layers].  Wide features (layer-wise kernels, one weight-gradient launch per layer): [Dense layers], [stage L], ...,
// [stage 1], [embed] -- L + 2 buckets, the all-reduce of each overlappable with the rest of the backward pass.
int n_phase_buckets(const v2x_model* m) { return m->F >= 128 ? m->L + 2 : 2; }
void phase_bucket_range(const v2x_model* m, int bucket, int64_t* b, int64_t* e) {
  if (m->F < 128) { *b = bucket_begin(m, bucket); *e = bucket_end(m, bucket); return; }
  if (bucket == 0) { *b = m->dense[0].off; *e = m->P; return; }
  const LayerDesc& ld = m->gnn[m->L + 1 - bucket];
  *b = ld.off; *e = ld.off + ld.slot_stride * m->S;
}

float adam_lr_t(const v2x_model* m, int64_t iteration) {
  const double t = (double)iteration;
  return (float)(m->cfg.lr * std::sqrt(1.0 - std::pow((double)m->cfg.beta2, t)) / (1.0 - std::pow((double)m->cfg.beta1, t)));
}

// fused_adam: the layers gnn[1..L] and dense[0] were updated by their weight-gradient launch (WideWgradArgs::adam): this
// launch covers what is left of the flat buffer, [0, gnn[1].off) and [dense[1].off, P)
int launch_reduce_adam(v2x_model* m, hipStream_t st, int n_slabs, bool do_adam, float* grad_dst, LossJob lj = LossJob{0, 0, 0, 0.f, 0},
                       int bucket = -1, bool fused_adam = false, int64_t range_begin = -1, int64_t range_end = -1,
                       bool advance_iteration = true) {
  AdamArgs a;
  memset(&a, 0, sizeof(a));
  a.param = m->params; a.grad = grad_dst ? grad_dst : m->grads; a.mom = m->mom; a.vel = m->vel;
  if (n_slabs > 0) {
    a.n_slabs = n_slabs; a.slab = m->slab; a.slab_stride = m->P;
    int nl = 0;
    for (auto* v : {&m->gnn, &m->dense})
      for (const LayerDesc& ld : *v) {
        a.layer_end4[nl] = (ld.off + ld.slot_stride * m->S) / 4;
        a.layer_slabs[nl] = ld.n_slabs;
        ++nl;
      }
    a.n_layers = nl;
  }
  a.n4 = m->P / 4;
  a.range_begin4 = bucket_begin(m, bucket) / 4; a.range_end4 = bucket_end(m, bucket) / 4;
  if (range_begin >= 0) { a.range_begin4 = range_begin / 4; a.range_end4 = range_end / 4; }
  if (fused_adam) {
    a.range_begin4 = 0; a.range_end4 = m->gnn[1].off / 4;
    a.range2_begin4 = m->dense[1].off / 4; a.range2_end4 = m->P / 4;
  }
  a.do_adam = do_adam ? 1 : 0;
  if (do_adam) {
    if (advance_iteration) m->iterations += 1;
    a.lr_t = adam_lr_t(m, m->iterations);
    a.beta1 = m->cfg.beta1; a.beta2 = m->cfg.beta2; a.eps = m->cfg.eps;
    if (m->pk_fwd) { a.pack.fwd = m->pk_fwd; a.pack.bwd = m->pk_bwd; a.pack.F = m->F; a.pack.S = m->S; a.pack.L = m->L; a.pack.xr = m->Dn + m->De; }
  }
  int max_slabs_used = 0;
  for (int l = 0; l < a.n_layers; ++l) max_slabs_used = std::max(max_slabs_used, a.layer_slabs[l]);
  a.groups = 1;
  if (a.slab && a.n4 < 256 * 1024 && max_slabs_used >= 16) a.groups = a.n4 < 64 * 1024 ? 16 : 4;
  const int cols = 256 / a.groups;
  int blocks = (int)((a.range_end4 - a.range_begin4 + a.range2_end4 - a.range2_begin4 + cols - 1) / cols);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  a.n_adam_blocks = blocks;
  a.rowloss = m->rowloss; a.loss = m->loss_dev; a.loss_n_idx = lj.n_idx; a.loss_stride = lj.stride; a.loss_scale = lj.scale; a.loss_slot_stride = lj.slot_stride;
  a.grad_direct = m->grads;
  a.loss_split = (lj.n_out == 1 && lj.n_idx > 16384) ? 64 : 1;
  a.loss_part = m->loss_part; a.loss_cnt = reinterpret_cast<unsigned*>(m->loss_part + 64);
  LAUNCH(m, do_adam ? (n_slabs > 0 ? "k_reduce_adam" : "k_adam") : "k_grad_reduce", k_reduce_adam,
         dim3(blocks + lj.n_out * a.loss_split), 0, st, a);
  return V2X_OK;
}

LossJob loss_job(const v2x_model* m, const DevBatch& d, int n_global) {
  const float inv = 1.0f / (float)((double)n_global * m->C);
  if (m->cfg.variable_graphs) return LossJob{1, d.R, 1, inv, 0};
  if (m->S == 1) return LossJob{m->N, d.B, m->N, inv, 1};          // shared weights: rowloss in node-row order
  return LossJob{m->N, d.B, 1, inv, srow_stride(m)};                // per-node weights: slot-major
}

// ------------------------------------------------------------------------------------ fused graph layers
// (kernels_fused.hpp) fixed-size graphs of <= 32 nodes, narrow features, no neighbour-init input, tile fits the LDS
int fused_rowf(int F) { return 4 * ((F / 16) | 1); }
size_t fused_lds(const v2x_model* m, const DevBatch& d, bool bwd, bool compl_sums = false) {
  const size_t rows = (size_t)FZ_TG * m->N;
  size_t b = 4 * rows * fused_rowf(m->F) * 4 + (rows + 1) * 4 + (size_t)FZ_TG * d.max_edges;
  b += rows * 4;                                       // bit masks: backward always, forward with compl_sums
  if (compl_sums) b += (size_t)(FZ_SUMS_ROWS + FZ_TOT_ROWS) * fused_rowf(m->F) * 4;
  return (b + 15) / 16 * 16 + 64;                      // + the turn flags of the backward (FzCtxB::sFlag)
}
bool fused_path(const v2x_model* m, const DevBatch& d) {
  if (!m->pk_fwd || m->cfg.variable_graphs || d.goff || d.nbr || m->F > 64 || m->N > 32 || m->L > FZ_MAXL) return false;
  if (d.max_nodes != m->N) return false;
  return fused_lds(m, d, true) <= 160 * 1024;
}
// Aggregations through the complement (kernels_fused.hpp, compl_sums): dense graphs whose extra LDS fits
bool fused_compl(const v2x_model* m, const DevBatch& d) {
  if (!m->compl_sums || m->N < 4) return false;
  if (2 * (int64_t)d.E <= (int64_t)d.B * m->N * (m->N - 1)) return false;      // average in-degree <= (N - 1) / 2
  return fused_lds(m, d, true, true) <= 160 * 1024;
}

__global__ void k_set4(float* dst, float a, float b, float c, float d) {
  if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
}
int launch_check(v2x_model* m, const char* kname) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) FAIL(m, V2X_EHIP, "launch %s failed: %s", kname, hipGetErrorString(e));
  return V2X_OK;
}

// Split tiles (kernels_fused_split.hpp): K workgroups per 16-graph tile when whole-tile workgroups would leave most of the
// chip idle (the shares of the metric's global batch).  All K x tiles workgroups must be co-resident (one per CU: their LDS
// tile admits no second one), and a member should keep at least four slots (below that the hand-overs cost more than the
// node updates they spread).  Edge form only.
int fused_split(const v2x_model* m, const DevBatch& d) {
  if (m->split_env == 0 || m->split_env == 1 || !fused_path(m, d) || m->L < 1 || m->L > 3) return 1;
  if (fused_lds(m, d, true) + (size_t)m->N * m->F * 4 > 160 * 1024) return 1;      // + the embed biases of all slots (forward)
  const int tiles = (d.B + FZ_TG - 1) / FZ_TG;
  auto fits = [&](int k) { return (m->N + k - 1) / k <= FZ_WAVES && tiles * k <= n_cus(); };      // one slot per wave, all co-resident
  if (m->split_env > 1) return (2 * m->split_env <= m->N && fits(m->split_env)) ? m->split_env : 1;
  for (int k : {5, 4, 3, 2})
    if (4 * k <= m->N && fits(k)) return k;
  return 1;
}
// The launch counters of a (re)allocated or re-armed exchange start at a value no counter of this process has had: tags must
// never repeat at an address, also not across buffers (a freed buffer's address is handed out again) or across a re-arm.
unsigned long long g_xchg_epoch_next = 1;
std::mutex g_xchg_epoch_mutex;                       // (models may be created from several host threads)
int arm_xchg(v2x_model* m, int tiles, unsigned long long floor) {
  unsigned long long base;
  {
    std::lock_guard<std::mutex> lock(g_xchg_epoch_mutex);
    g_xchg_epoch_next = std::max(g_xchg_epoch_next, floor);
    base = g_xchg_epoch_next;
    g_xchg_epoch_next += 1ull << 16;
  }
  std::vector<unsigned long long> init((size_t)tiles, base);
  HIPCHK(m, hipMemset(m->xchg_buf, 0, xchg_bytes(m, tiles)));                 // tag 0 = never written
  HIPCHK(m, hipMemcpy(m->xchg_sync, init.data(), (size_t)tiles * sizeof(unsigned long long), hipMemcpyHostToDevice));
  HIPCHK(m, hipDeviceSynchronize());
  return V2X_OK;
}
int xchg_max_count(v2x_model* m, unsigned long long* out) {
  *out = 0;
  if (!m->xchg_sync || m->xchg_cap_tiles <= 0) return V2X_OK;
  std::vector<unsigned long long> cur((size_t)m->xchg_cap_tiles);
  HIPCHK(m, hipMemcpy(cur.data(), m->xchg_sync, cur.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (unsigned long long v : cur) *out = std::max(*out, v);
  return V2X_OK;
}
int ensure_xchg(v2x_model* m, int tiles) {
  if (tiles <= m->xchg_cap_tiles) return V2X_OK;
  if (m->capturing) FAIL(m, V2X_ESTATE, "exchange buffer growth during graph capture");
  drop_graphs(m);
  HIPCHK(m, hipDeviceSynchronize());
  unsigned long long seen = 0;
  CHK(xchg_max_count(m, &seen));
  if (m->xchg_buf) HIPCHK(m, hipFree(m->xchg_buf));
  if (m->xchg_sync) HIPCHK(m, hipFree(m->xchg_sync));
  m->xchg_buf = nullptr; m->xchg_sync = nullptr; m->xchg_cap_tiles = 0;
  HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&m->xchg_buf), xchg_bytes(m, tiles)));
  HIPCHK(m, hipMalloc(reinterpret_cast<void**>(&m->xchg_sync), (size_t)tiles * sizeof(unsigned long long)));
  CHK(arm_xchg(m, tiles, seen + 1));
  m->xchg_cap_tiles = tiles;
  return V2X_OK;
}
FzXchg xchg_args(const v2x_model* m, int K) { return FzXchg{m->xchg_buf, m->xchg_sync, m->xchg_cap_tiles, K}; }

int launch_pack(v2x_model* m, hipStream_t st) {
  PackArgs p;
  memset(&p, 0, sizeof(p));
  p.params = m->params; p.pk_fwd = m->pk_fwd; p.pk_bwd = m->pk_bwd; p.S = m->S;
  for (int s = 0; s <= m->L; ++s) { p.layer_off[s] = m->gnn[s].off; p.slot_stride[s] = m->gnn[s].slot_stride; p.pad[s] = m->gnn[s].pad; }
  const dim3 grid(4, m->S, m->L + 1);
  switch (m->F) {
    case 16: { auto k = k_pack_weights<16>; LAUNCH(m, "k_pack_weights", k, grid, 0, st, p); break; }
    case 32: { auto k = k_pack_weights<32>; LAUNCH(m, "k_pack_weights", k, grid, 0, st, p); break; }
    case 64: { auto k = k_pack_weights<64>; LAUNCH(m, "k_pack_weights", k, grid, 0, st, p); break; }
    default: FAIL(m, V2X_EINVAL, "pack: unsupported feat_dim %d", m->F);
  }
  return V2X_OK;
}

#define LAUNCH_T(m, kname, kern, grid, threads, lds, stream, ...)                         \
  do {                                                                                  \
    ProfRec _r;                                                                         \
    const bool _p = (m) && (m)->prof && !(m)->capturing;                                \
    if (_p) {                                                                           \
      _r.id = prof_id(m, kname);                                                          \
      hipEventCreate(&_r.ev0); hipEventCreate(&_r.ev1);                                     \
      hipEventRecord(_r.ev0, stream);                                                     \
    }                                                                                   \
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stream, __VA_ARGS__);            \
    if (_p) { hipEventRecord(_r.ev1, stream); (m)->prof_recs.push_back(_r); }             \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) FAIL(m, V2X_EHIP, "launch %s failed: %s", kname, hipGetErrorString(_e)); \
  } while (0)

// Fragment-major hand-off between the fused graph-layer kernels and k_mlp_train_wg (MlpArgs::frag_groups): whole batch in
// whole 16-graph groups, per-node weights (a tile of the MLP kernel is then a group of one node), at least one graph layer.
bool mlp_wg_path(const v2x_model* m);
bool dense0_rides(const v2x_model* m, const DevBatch& d, const IdxMap& x);
bool frag_layout(const v2x_model* m, const DevBatch& d, Range r) {
  static const int on = env_int("V2X_FRAG_HANDOFF", 1), per_stage = env_int("V2X_WG_PER_STAGE", 0);
  // (V2X_FRAG_WITH_DENSE0_ROLE=0: row-major hand-over where Dense-0's weight gradient is a role of k_wgrad -- measured slower: the
  //  fragment-major reader of that role costs nothing next to what the MLP launch and the fused kernels gain from 1 KiB accesses,
  //  0.1131 against 0.1157 ms at the 512-graph share)
  static const int frag_d0 = env_int("V2X_FRAG_WITH_DENSE0_ROLE", 1);
  return on && !per_stage && fused_path(m, d) && r.g0 == 0 && r.ng == d.B && mlp_wg_path(m) && m->S == m->N && m->L >= 1 &&
         d.B % FZ_TG == 0 && (frag_d0 || !dense0_rides(m, d, idx_map(m, d, r)));
}

int launch_fused_fwd(v2x_model* m, hipStream_t st, const DevBatch& d, bool frag_out = false) {
  static const int split_fwd = env_int("V2X_FUSED_SPLIT_FWD", 1);      // (debugging: one direction on whole tiles)
  // The copy follows the parameters by itself: Adam writes both (k_reduce_adam / pack_scatter), set / copy_weights
  // re-pack eagerly.  Only a caller that took the raw parameter pointer (v2x_param_ptr) forces a re-pack per forward.
  if (m->pk_stale) { CHK(launch_pack(m, st)); m->pk_stale = m->raw_params; }
  FusedFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.xe = d.xe; a.row_ptr = d.rp; a.col_idx = d.ci; a.pk = m->pk_fwd;
  for (int s = 0; s <= m->L; ++s) { a.h[s] = m->h[s]; a.a[s] = m->a[s]; a.gate[s] = m->gate_bits + s * m->gate_stride; }
  a.n_graphs = d.B; a.N = m->N; a.L = m->L; a.S = m->S; a.edges_cap = FZ_TG * d.max_edges; a.n_edges = d.E; a.err = m->flag_dev;
  a.frag_out = frag_out ? 1 : 0;
  a.nbmask = m->nbmask;
  const int tiles = (d.B + FZ_TG - 1) / FZ_TG;
  if (const int K = fused_split(m, d); K > 1 && split_fwd) {            // the shares of the global batch: K workgroups per tile
    CHK(ensure_xchg(m, tiles));
    const FzXchg xg = xchg_args(m, K);
    a.ts = m->ts_buf;
    const dim3 grid((tiles + 7) / 8 * 8 * K);
    const size_t lds = fused_lds(m, d, false, false) + (size_t)m->N * m->F * 4;
    ProfRec r;
    const bool pr = m->prof && !m->capturing;
    if (pr) { r.id = prof_id(m, "k_gnn_fwd_fused"); hipEventCreate(&r.ev0); hipEventCreate(&r.ev1); hipEventRecord(r.ev0, st); }
    int rc = V2X_EINVAL;
    do {
      if (m->ts_buf && m->F == 64 && m->L == 2) { hipLaunchKernelGGL((k_gnn_fwd_split<64, 2, true>), grid, dim3(FZ_THREADS), lds, st, a, xg); rc = launch_check(m, "k_gnn_fwd_split"); break; }
#define V2X_FZ_FWDS(FF, LL) if (m->F == FF && m->L == LL) { hipLaunchKernelGGL((k_gnn_fwd_split<FF, LL>), grid, dim3(FZ_THREADS), lds, st, a, xg); rc = launch_check(m, "k_gnn_fwd_split"); break; }
      V2X_FZ_FWDS(16, 1) V2X_FZ_FWDS(16, 2) V2X_FZ_FWDS(16, 3) V2X_FZ_FWDS(32, 1) V2X_FZ_FWDS(32, 2) V2X_FZ_FWDS(32, 3) V2X_FZ_FWDS(64, 1) V2X_FZ_FWDS(64, 2) V2X_FZ_FWDS(64, 3)
#undef V2X_FZ_FWDS
    } while (0);
    if (pr) { hipEventRecord(r.ev1, st); m->prof_recs.push_back(r); }
    if (rc == V2X_EINVAL && m->err.empty()) FAIL(m, V2X_EINVAL, "split fused forward: unsupported shape");
    return rc;
  }
  const dim3 grid(tiles);
  // (a batch the split-tile kernels are picked for runs the edge form in BOTH directions, also when a debugging switch keeps one
  //  of them on whole tiles: the two kernels hand the rows' IN-neighbour sets over, the complement form would read them as
  //  non-neighbour sets)
  a.compl_sums = (fused_split(m, d) <= 1 && fused_compl(m, d)) ? 1 : 0;
  const size_t lds = fused_lds(m, d, false, a.compl_sums != 0);
  const int spw = (m->N + FZ_WAVES - 1) / FZ_WAVES;
#define V2X_FZ_FWD(FF, SP)                                                                                            \
  if (m->F == FF && spw == SP) {                                                                                      \
    if (FF == 64 && SP == 3 && m->ts_buf) {                                                                          \
      a.ts = m->ts_buf;                                                                                               \
      if (a.compl_sums) { auto k = k_gnn_fwd_fused<64, 3, true, true>; LAUNCH_T(m, "k_gnn_fwd_fused", k, grid, FZ_THREADS, lds, st, a); } \
      else { auto k = k_gnn_fwd_fused<64, 3, true>; LAUNCH_T(m, "k_gnn_fwd_fused", k, grid, FZ_THREADS, lds, st, a); } \
    } else if (a.compl_sums) { auto k = k_gnn_fwd_fused<FF, SP, false, true>; LAUNCH_T(m, "k_gnn_fwd_fused", k, grid, FZ_THREADS, lds, st, a); } \
    else { auto k = k_gnn_fwd_fused<FF, SP>; LAUNCH_T(m, "k_gnn_fwd_fused", k, grid, FZ_THREADS, lds, st, a); }        \
    return V2X_OK;                                                                                                    \
  }
  V2X_FZ_FWD(16, 1) V2X_FZ_FWD(16, 2) V2X_FZ_FWD(16, 3) V2X_FZ_FWD(16, 4)
  V2X_FZ_FWD(32, 1) V2X_FZ_FWD(32, 2) V2X_FZ_FWD(32, 3) V2X_FZ_FWD(32, 4)
  V2X_FZ_FWD(64, 1) V2X_FZ_FWD(64, 2) V2X_FZ_FWD(64, 3) V2X_FZ_FWD(64, 4)
#undef V2X_FZ_FWD
  FAIL(m, V2X_EINVAL, "fused forward: unsupported shape");
}

int launch_fused_bwd(v2x_model* m, hipStream_t st, const DevBatch& d) {
  static const int split_bwd = env_int("V2X_FUSED_SPLIT_BWD", 1);      // (debugging: one direction on whole tiles)
  FusedBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.row_ptr = d.rp; a.col_idx = d.ci; a.pk = m->pk_bwd; a.gha = m->gha;
  for (int s = 0; s <= m->L; ++s) a.dpre[s] = m->dpre[s];
  for (int s = 0; s < m->L; ++s) a.gate[s] = m->gate_bits + s * m->gate_stride;
  a.n_graphs = d.B; a.N = m->N; a.L = m->L; a.S = m->S; a.edges_cap = FZ_TG * d.max_edges; a.n_edges = d.E; a.err = m->flag_dev;
  a.frag_gha = m->frag_live ? 1 : 0;
  a.nbmask = m->nbmask;
  const int tiles = (d.B + FZ_TG - 1) / FZ_TG;
  if (const int K = fused_split(m, d); K > 1 && split_bwd) {
    CHK(ensure_xchg(m, tiles));
    const FzXchg xg = xchg_args(m, K);
    a.ts = m->ts_buf;
    const dim3 grid((tiles + 7) / 8 * 8 * K);
    const size_t lds = fused_lds(m, d, true, false);
    ProfRec r;
    const bool pr = m->prof && !m->capturing;
    if (pr) { r.id = prof_id(m, "k_gnn_bwd_fused"); hipEventCreate(&r.ev0); hipEventCreate(&r.ev1); hipEventRecord(r.ev0, st); }
    int rc = V2X_EINVAL;
    do {
      if (m->ts_buf && m->F == 64 && m->L == 2) { hipLaunchKernelGGL((k_gnn_bwd_split<64, 2, true>), grid, dim3(FZ_THREADS), lds, st, a, xg); rc = launch_check(m, "k_gnn_bwd_split"); break; }
#define V2X_FZ_BWDS(FF, LL) if (m->F == FF && m->L == LL) { hipLaunchKernelGGL((k_gnn_bwd_split<FF, LL>), grid, dim3(FZ_THREADS), lds, st, a, xg); rc = launch_check(m, "k_gnn_bwd_split"); break; }
      V2X_FZ_BWDS(16, 1) V2X_FZ_BWDS(16, 2) V2X_FZ_BWDS(16, 3) V2X_FZ_BWDS(32, 1) V2X_FZ_BWDS(32, 2) V2X_FZ_BWDS(32, 3) V2X_FZ_BWDS(64, 1) V2X_FZ_BWDS(64, 2) V2X_FZ_BWDS(64, 3)
#undef V2X_FZ_BWDS
    } while (0);
    if (pr) { hipEventRecord(r.ev1, st); m->prof_recs.push_back(r); }
    if (rc == V2X_EINVAL && m->err.empty()) FAIL(m, V2X_EINVAL, "split fused backward: unsupported shape");
    return rc;
  }
  const dim3 grid(tiles);
  // (a batch the split-tile kernels are picked for runs the edge form in BOTH directions, also when a debugging switch keeps one
  //  of them on whole tiles: the two kernels hand the rows' IN-neighbour sets over, the complement form would read them as
  //  non-neighbour sets)
  a.compl_sums = (fused_split(m, d) <= 1 && fused_compl(m, d)) ? 1 : 0;
  const size_t lds = fused_lds(m, d, true, a.compl_sums != 0);
  const int spw = (m->N + FZ_WAVES - 1) / FZ_WAVES;
#define V2X_FZ_BWD(FF, SP)                                                                                            \
  if (m->F == FF && spw == SP) {                                                                                      \
    if (FF == 64 && SP == 3 && m->ts_buf) {                                                                          \
      a.ts = m->ts_buf;                                                                                               \
      if (a.compl_sums) { auto k = k_gnn_bwd_fused<64, 3, true, true>; LAUNCH_T(m, "k_gnn_bwd_fused", k, grid, FZ_THREADS, lds, st, a); } \
      else { auto k = k_gnn_bwd_fused<64, 3, true>; LAUNCH_T(m, "k_gnn_bwd_fused", k, grid, FZ_THREADS, lds, st, a); } \
    } else if (a.compl_sums) { auto k = k_gnn_bwd_fused<FF, SP, false, true>; LAUNCH_T(m, "k_gnn_bwd_fused", k, grid, FZ_THREADS, lds, st, a); } \
    else { auto k = k_gnn_bwd_fused<FF, SP>; LAUNCH_T(m, "k_gnn_bwd_fused", k, grid, FZ_THREADS, lds, st, a); }        \
    return V2X_OK;                                                                                                    \
  }
  V2X_FZ_BWD(16, 1) V2X_FZ_BWD(16, 2) V2X_FZ_BWD(16, 3) V2X_FZ_BWD(16, 4)
  V2X_FZ_BWD(32, 1) V2X_FZ_BWD(32, 2) V2X_FZ_BWD(32, 3) V2X_FZ_BWD(32, 4)
  V2X_FZ_BWD(64, 1) V2X_FZ_BWD(64, 2) V2X_FZ_BWD(64, 3) V2X_FZ_BWD(64, 4)
#undef V2X_FZ_BWD
  FAIL(m, V2X_EINVAL, "fused backward: unsupported shape");
}

// ------------------------------------------------------------------------------------ passes
// ------------------------------------------------------------------------------------ few-graph predict, one launch
constexpr int SMALL_ROWS = 256;             // node rows (= workgroups) of one launch: all co-resident on any gfx950 part
bool small_path(const v2x_model* m, const DevBatch& d) {
  if (!m->small_predict || !m->small_h || m->cfg.variable_graphs || d.goff || d.nbr || m->F > 64 || m->L > FZ_MAXL) return false;
  if (m->N > 32) return false;                 // the kernel keeps a node's in-neighbours in a 32-entry LDS list
  return d.max_nodes == m->N && d.R <= std::min(SMALL_ROWS, n_cus());
}

constexpr size_t PIN_XE = 0, PIN_RP = 16384, PIN_CI = 20480, PIN_Q = 53248, PIN_BYTES = 65536;   // SMALL_ROWS rows, <= 31 in-edges each

int launch_small_forward(v2x_model* m, hipStream_t st, const DevBatch& d, float* q_dst = nullptr) {
  SmallFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.xe = d.xe; a.row_ptr = d.rp; a.col_idx = d.ci; a.params = m->params;
  for (int s = 0; s <= m->L; ++s) { a.gnn_off[s] = m->gnn[s].off; a.gnn_sstride[s] = m->gnn[s].slot_stride; }
  for (int i = 0; i < 4; ++i) { a.dense_off[i] = m->dense[i].off; a.dense_sstride[i] = m->dense[i].slot_stride; }
  a.hbuf = m->small_h; a.sync = m->small_sync; a.err = m->flag_dev; a.q = q_dst ? q_dst : m->q;
  a.N = m->N; a.L = m->L; a.S = m->S; a.C = m->C; a.Dn = m->Dn; a.De = m->De; a.n_rows = d.R; a.slab_rows = SMALL_ROWS;
  const dim3 grid(m->N, d.B);
#define V2X_SMALL(FF)                                                                                   \
  if (m->F == FF) {                                                                                     \
    auto k = k_predict_small<FF>;                                                                       \
    LAUNCH_T(m, "k_predict_small", k, grid, SM_BLOCK, 0, st, a);                                        \
    return V2X_OK;                                                                                      \
  }
  V2X_SMALL(16) V2X_SMALL(32) V2X_SMALL(64)
#undef V2X_SMALL
  FAIL(m, V2X_EINVAL, "small forward: unsupported feat_dim %d", m->F);
}

// ------------------------------------------------------------------------------------ ragged fused forward
// (kernels_ragged.hpp) variable-size graphs of <= 128 nodes, shared weights, narrow features, dense enough for the bit masks
bool ragged_fused_path(const v2x_model* m, const DevBatch& d) {
  // (any density: a row's aggregation walks its mask's one bits or -- rows with more edges than non-edges in graphs of >= 16
  //  nodes -- the zero bits; the masks are built for this path whether or not the dense MFMA aggregation wants them)
  return m->ragged_fused && m->cfg.variable_graphs && m->S == 1 && d.goff && !d.nbr && m->F <= 64 && m->L <= FZ_MAXL && d.max_nodes <= 128 &&
         d.max_nodes <= RG_CAP / 2;
}
bool need_adj_masks(const v2x_model* m, const DevBatch& d) { return use_dense_agg(d, m->F) || ragged_fused_path(m, d); }
// Small tiles (kernels_ragged_small.hpp: 160-row tiles, 4-wave workgroups, weights as fragments from L2, three workgroups per
// CU): built and measured in round 5 -- forward 123 us against 101, backward 161 against 103 at configs[4]'s share, + 19 us for
// the plan as a launch of its own (profiles/r05_ragged_small_ab.txt) -- so OFF unless V2X_RAGGED_SMALL=1 (read at create too:
// the fragment-major copy of a ragged model's weights is only kept then)
bool ragged_small(const v2x_model* m, const DevBatch& d) {
  return m->ragged_small_env && m->pk_fwd && m->pk_bwd && d.max_nodes <= 128 && m->L >= 1;
}
int ragged_cap(const v2x_model* m, const DevBatch& d) { return ragged_small(m, d) ? RGS_CAP : RG_CAP; }
int ragged_capp(const v2x_model* m, const DevBatch& d) { return ragged_cap(m, d) - d.max_nodes + 1; }
int ragged_wgs(const v2x_model* m, const DevBatch& d) { return (d.R + ragged_capp(m, d) - 1) / ragged_capp(m, d); }

// k_ragged_plan's tables in LDS; past that (tens of thousands of graphs in one batch) the interval plan of k_adj_masks
int ragged_plan_words(const v2x_model* m, const DevBatch& d) { return 3 * (d.B + 1) + ragged_wgs(m, d) + 1; }
bool ragged_packed_plan(const v2x_model* m, const DevBatch& d) { return m->ragged_packed && ragged_plan_words(m, d) <= RG_PLAN_LDS_WORDS; }

template <int F>
int launch_ragged_fwd_f(v2x_model* m, hipStream_t st, const RaggedFwdArgs& a, int n_wgs) {
  auto k = k_gnn_fwd_ragged<F>;
  static const bool once = [] { allow_big_lds((const void*)k_gnn_fwd_ragged<F>); return true; }();
  (void)once;
  LAUNCH_T(m, "k_gnn_fwd_ragged", k, dim3(n_wgs), RG_THREADS, (size_t)RaggedLds<F>::TOTAL * 4, st, a);
  return V2X_OK;
}

template <int F, bool BWD, typename Args>
int launch_ragged_small_f(v2x_model* m, hipStream_t st, const Args& a, int n_wgs) {
  if constexpr (BWD) {
    auto k = k_gnn_bwd_ragged_s<F>;
    LAUNCH_T(m, "k_gnn_bwd_ragged", k, dim3(n_wgs), RGS_THREADS, (size_t)RaggedSmallBwdLds<F>::TOTAL * 4, st, a);
  } else {
    auto k = k_gnn_fwd_ragged_s<F>;
    LAUNCH_T(m, "k_gnn_fwd_ragged", k, dim3(n_wgs), RGS_THREADS, (size_t)RaggedSmallLds<F>::TOTAL * 4, st, a);
  }
  return V2X_OK;
}

int launch_ragged_fwd(v2x_model* m, hipStream_t st, const DevBatch& d) {
  RaggedFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.xe = d.xe; a.graph_off = d.goff; a.row_ptr = d.rp;
  a.mask_words = (d.max_nodes + 31) / 32;
  a.adjT = (const unsigned*)m->adj_mask.p + (size_t)d.R * a.mask_words;
  a.plan = (const int32_t*)m->plan_buf.p;
  for (int s = 0; s <= m->L; ++s) { a.W[s] = m->params + m->gnn[s].off; a.h[s] = m->h[s]; a.a[s] = m->a[s]; }
  a.n_graphs = d.B; a.n_rows = d.R; a.L = m->L; a.capp = ragged_capp(m, d); a.xr = m->Dn + m->De; a.err = m->flag_dev;
  a.ts = m->ts_buf;
  const int n_wgs = ragged_wgs(m, d);
  if (ragged_small(m, d)) {                             // 160-row tiles, weights as fragments from L2 (kernels_ragged_small.hpp)
    if (m->pk_stale) { CHK(launch_pack(m, st)); m->pk_stale = m->raw_params; }
    RaggedSmallFwdArgs sa{a, m->pk_fwd};
    switch (m->F) {
      case 16: return launch_ragged_small_f<16, false>(m, st, sa, n_wgs);
      case 32: return launch_ragged_small_f<32, false>(m, st, sa, n_wgs);
      case 64: return launch_ragged_small_f<64, false>(m, st, sa, n_wgs);
    }
  }
  switch (m->F) {
    case 16: return launch_ragged_fwd_f<16>(m, st, a, n_wgs);
    case 32: return launch_ragged_fwd_f<32>(m, st, a, n_wgs);
    case 64: return launch_ragged_fwd_f<64>(m, st, a, n_wgs);
  }
  FAIL(m, V2X_EINVAL, "ragged forward: unsupported feat_dim %d", m->F);
}

template <int F>
int launch_ragged_bwd_f(v2x_model* m, hipStream_t st, const RaggedBwdArgs& a, int n_wgs) {
  auto k = k_gnn_bwd_ragged<F>;
  static const bool once = [] { allow_big_lds((const void*)k_gnn_bwd_ragged<F>); return true; }();
  (void)once;
  LAUNCH_T(m, "k_gnn_bwd_ragged", k, dim3(n_wgs), RG_THREADS, (size_t)RaggedBwdLds<F>::TOTAL * 4, st, a);
  return V2X_OK;
}

// L + 1 transposed aggregations + L data gradients of ragged graphs in one launch (masks and plan: the forward's)
int launch_ragged_bwd(v2x_model* m, hipStream_t st, const DevBatch& d) {
  RaggedBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.gha = m->gha; a.graph_off = d.goff;
  a.mask_words = (d.max_nodes + 31) / 32;
  a.adj = (const unsigned*)m->adj_mask.p;
  a.plan = (const int32_t*)m->plan_buf.p;
  for (int s = 0; s <= m->L; ++s) { a.W[s] = m->params + m->gnn[s].off; a.h[s] = m->h[s]; a.dpre[s] = m->dpre[s]; }
  a.n_graphs = d.B; a.n_rows = d.R; a.L = m->L; a.capp = ragged_capp(m, d); a.xr = m->Dn + m->De; a.err = m->flag_dev;
  const int n_wgs = ragged_wgs(m, d);
  if (ragged_small(m, d)) {
    RaggedSmallBwdArgs sa{a, m->pk_bwd};
    switch (m->F) {
      case 16: return launch_ragged_small_f<16, true>(m, st, sa, n_wgs);
      case 32: return launch_ragged_small_f<32, true>(m, st, sa, n_wgs);
      case 64: return launch_ragged_small_f<64, true>(m, st, sa, n_wgs);
    }
  }
  switch (m->F) {
    case 16: return launch_ragged_bwd_f<16>(m, st, a, n_wgs);
    case 32: return launch_ragged_bwd_f<32>(m, st, a, n_wgs);
    case 64: return launch_ragged_bwd_f<64>(m, st, a, n_wgs);
  }
  FAIL(m, V2X_EINVAL, "ragged backward: unsupported feat_dim %d", m->F);
}

// OPT-IN (V2X_MLP_STREAM=1; measured slower, profiles/r06_mlp_stream_ab.txt): the decision MLP as one wave per (slot, tile) with its
// weights streamed from L2 (kernels_mlpstream.hpp), ALL four Dense weight gradients as roles of the graph layers' launch -- wherever
// Dense-0's weight gradient may ride (dense0_rides).  At the 512- / 1024- / 2048-graph shares the MLP launch itself goes 23.6 -> 19.1,
// 35.3 -> 33.0, 48.2 -> 50.9 us (every wave of a slot streams the same 135 KB in the same order: 4.5-6.9 TB/s over the chip, 1.6-1.9x
// the MFMA time even at 2.5 waves per SIMD), the Dense 1..3 roles chained behind the Dense-0 halves add 10-27 us to the
// weight-gradient launch (three more operand pipelines and accumulator exchanges per workgroup) and the image 13.6 us per step
// (it could ride on k_reduce_adam's scatter): 0.1284 / 0.1696 / 0.2274 ms per step against 0.1128 / 0.1445 / 0.1870.
bool mlp_stream_rides(const v2x_model* m, const DevBatch& d, const IdxMap& x) {
  if (env_int("V2X_MLP_STREAM", 0) != 1) return false;             // (read per call: the tests switch it inside one process)
  if (!m->mlp_img || m->dqn_tq || m->S != m->N || m->L + 5 > WG_MAX_ROLES) return false;
  return env_int("V2X_MLP_WG0", -1) != 1 && dense0_rides(m, d, x);
}

int launch_mlp_stream(v2x_model* m, hipStream_t st, const MlpArgs& a, int n_slots) {
  // the images follow the parameters: rebuilt from the flat buffer in front of every use (1.5 MB, one workgroup per slot)
  { auto k = k_mlp_image<64>; LAUNCH_T(m, "k_mlp_image", k, dim3(n_slots), 256, (size_t)MlpLds<64>::TOTAL * 4, st, a, m->mlp_img); }
  const int T = (a.n_idx + 15) / 16, units = T * n_slots;
  if (a.frag_groups > 0) { auto k = k_mlp_stream<64, true>; LAUNCH_T(m, "k_mlp_stream", k, dim3(units), 64, 0, st, a, (const float*)m->mlp_img, T, units); }
  else { auto k = k_mlp_stream<64, false>; LAUNCH_T(m, "k_mlp_stream", k, dim3(units), 64, 0, st, a, (const float*)m->mlp_img, T, units); }
  return V2X_OK;
}

int run_forward(v2x_model* m, hipStream_t st, const DevBatch& d, Range r, bool with_mlp = true) {
  const int F = m->F, L = m->L;
  const IdxMap x = idx_map(m, d, r);
  m->frag_live = !with_mlp && frag_layout(m, d, r);
  const bool ragged = ragged_fused_path(m, d) && r.g0 == 0 && r.ng == d.B;
  if (fused_path(m, d) && r.g0 == 0 && r.ng == d.B) {
    CHK(launch_fused_fwd(m, st, d, m->frag_live));       // embed + L stages + L+1 aggregations: one launch
  } else {
    if (need_adj_masks(m, d)) {
      AggDenseArgs q = agg_dense_args(d, r, m->N, F);
      q.adj = (unsigned*)m->adj_mask.p;
      q.adjT = q.adj + (size_t)d.R * q.mask_words;
      q.err = m->flag_dev;
      const bool packed = ragged && ragged_packed_plan(m, d);
      if (ragged) m->plan_len = ragged_wgs(m, d) + 1;
      if (ragged && !packed) { q.plan = (int32_t*)m->plan_buf.p; q.plan_capp = ragged_capp(m, d); q.plan_n = ragged_wgs(m, d); }
      // the packed plan by workgroup 0 of the mask launch while its offsets + 16-bit tables leave the mask workgroups their
      // eight per CU (<= 19 KB of LDS: ~2,200 graphs); a launch of its own (k_ragged_plan, 32-bit tables) beyond
      const bool folded = packed && m->ragged_plan_fold && d.B < 65000 && plan16_bytes(d.B, ragged_wgs(m, d)) <= 19 * 1024;
      if (folded) { q.plan = (int32_t*)m->plan_buf.p; q.plan_cap = ragged_cap(m, d); q.plan_n = ragged_wgs(m, d); }
      // (the plan reads the offsets, the masks the CSR: independent -- but as a forked branch of the captured step the two
      //  cost MORE than one after the other: 0.466 against 0.455 ms per configs[4] step, the graph's cross-branch
      //  dependencies outweigh the 5 us the plan takes)
      if (packed && !folded) {
        static const bool once = [] { allow_big_lds((const void*)k_ragged_plan); return true; }();
        (void)once;
        RaggedPlanArgs pa{d.goff, (int32_t*)m->plan_buf.p, d.B, ragged_wgs(m, d), ragged_cap(m, d)};
        LAUNCH_T(m, "k_ragged_plan", k_ragged_plan, dim3(1), RG_PLAN_THREADS, (size_t)ragged_plan_words(m, d) * 4, st, pa);
      }
      CHK(build_adj_masks(m, st, q));
    }
    if (ragged) {                                        // embed + L stages + L + 1 aggregations of ragged graphs: one launch
      CHK(launch_ragged_fwd(m, st, d));
      goto mlp;
    }
    CHK(launch_node_fwd(m, st, 0, x, d.xe, nullptr, d.nbr, m->h[0]));
    CHK(launch_agg(m, st, d, r, m->N, F, m->h[0], F, nullptr, 0, nullptr, m->a[0], 0));
    for (int s = 1; s <= L; ++s) {
      CHK(launch_node_fwd(m, st, s, x, d.xe, m->h[s - 1], m->a[s - 1], m->h[s]));
      CHK(launch_agg(m, st, d, r, m->N, F, m->h[s], F, nullptr, 0, nullptr, m->a[s], 0));
    }
  }
mlp:
  if (!with_mlp) return V2X_OK;          // training: the MLP runs fused with its backward (k_mlp_train)
  MlpArgs a;
  mlp_args(m, a, x, d.xe, m->h[L], m->a[L]);
  CHK(launch_mlp(m, st, a, false));
  return V2X_OK;
}

float loss_denominator(const v2x_model* m, int n_global) {
  // fixed-N: per-output mean over (B_global, C)  (tf.losses.huber_loss per output, BS_brain.py:214)
  // variable graphs: n_global counts node rows; single mean over (rows_global, C)
  return (float)((double)n_global * m->C);
}

// backward kernel chain of one range.  sw != st: the weight-gradient kernels go to the side stream `sw`,
// forked after the kernel producing their dpre (they only consume saved activations + dpre_s, and nothing
// consumes them before the slab reduction); the caller joins sw back.
int run_backward(v2x_model* m, hipStream_t st, hipStream_t sw, const DevBatch& d, Range r, const float* y_dev,
                 int n_global) {
  const int F = m->F, L = m->L;
  const IdxMap x = idx_map(m, d, r);
  const bool two = sw != st;
  int evi = 0;
  auto fork = [&]() -> int {      // side stream waits for everything issued on st so far
    if (!two) return V2X_OK;
    hipEvent_t e = m->ev[evi++];
    HIPCHK(m, hipEventRecord(e, st));
    HIPCHK(m, hipStreamWaitEvent(sw, e, 0));
    return V2X_OK;
  };
  MlpArgs a;
  mlp_args(m, a, x, d.xe, m->h[L], m->a[L]);
  a.y = y_dev;
  a.inv_denom = 1.0f / loss_denominator(m, n_global);
  const bool mlp_wg = mlp_wg_path(m);      // the Dense weight gradients come out of the MLP launch itself
  // wide path: every layer's weight gradient in ONE launch after the data chain -- unless somebody wants a layer's gradient
  // as soon as it is final (per-layer all-reduce buckets of data parallelism: m->bucketed)
  static const int wide_merge = env_int("V2X_WIDE_MERGE", 1);
  m->wide_merge_now = is_wide(m) && wide_merge && !two && !m->bucketed && m->L + 2 <= WWM_ROLES;
  struct MergeGuard { v2x_model* m; ~MergeGuard() { m->wide_merge_now = false; } } merge_guard{m};
  if (m->frag_live && !frag_layout(m, d, r)) FAIL(m, V2X_ESTATE, "backward: the saved forward is fragment-major, this backward cannot read it");
  a.frag_groups = m->frag_live ? d.B / FZ_TG : 0;
  if (m->dqn_tq) {
    if (!mlp_wg) FAIL(m, V2X_ESTATE, "backward: in-kernel DQN targets need the k_mlp_train_wg path");
    a.tq = m->dqn_tq; a.y = m->dqn_tq; a.q = m->dqn_y;
  }
  // (the weight-gradient launch below must be the merged one of wgrad_gnn_all: one stream, no per-stage split)
  m->dense0_out_now = mlp_wg && !two && r.g0 == 0 && r.ng == d.B && dense0_rides(m, d, x);
  struct D0Guard { v2x_model* m; ~D0Guard() { m->dense0_out_now = false; } } d0_guard{m};
  m->mlp_stream_now = m->dense0_out_now && mlp_stream_rides(m, d, x);
  struct MsGuard { v2x_model* m; ~MsGuard() { m->mlp_stream_now = false; } } ms_guard{m};
  if (m->mlp_stream_now) CHK(launch_mlp_stream(m, st, a, x.grid_y));
  else if (mlp_wg) CHK(launch_mlp_train_wg(m, st, a, !m->dense0_out_now));
  else if (mlp_fused_training(m)) CHK(launch_mlp_train(m, st, a));
  else CHK(launch_mlp(m, st, a, true));
  const bool merged = !mlp_wg && !two && wgrad_all_fits(m) && env_int("V2X_WG_SPLIT", 0) == 0;
  CHK(fork());
  if (!merged && !mlp_wg) CHK(wgrad_mlp(m, sw, x, d.xe, m->h[L], m->a[L]));        // 4 Dense layers, one launch (side stream if two)
  // V2X_WG_PER_STAGE=1: every GNN stage's weight gradient goes to the side stream as soon as its dpre exists
  // (overlaps the remaining agg/dgrad chain) instead of one fused launch after the chain
  static const bool per_stage = env_int("V2X_WG_PER_STAGE", 0) != 0;
  const bool split = two && per_stage && !is_wide(m);
  const bool fused = fused_path(m, d) && r.g0 == 0 && r.ng == d.B && !split;
  const bool ragged_bwd = m->ragged_fused_bwd;
  if (fused) {
    CHK(launch_fused_bwd(m, st, d));       // L+1 transposed aggregations + L data gradients: one launch
  } else if (ragged_bwd && !split && ragged_fused_path(m, d) && r.g0 == 0 && r.ng == d.B) {
    CHK(launch_ragged_bwd(m, st, d));      // the same for ragged graphs (masks and plan from this step's forward)
  } else {
    for (int s = L; s >= 1; --s) {
      // dpre_s = (dh_direct + Agg^T(dagg)) * relu'(h_s)
      CHK(launch_agg(m, st, d, r, m->N, F, m->gha + F, 2 * F, m->gha, 2 * F, s < L ? m->h[s] : nullptr, m->dpre[s], 1));
      if (split) { CHK(fork()); CHK(wgrad_gnn(m, sw, s, x, d.xe, m->h[s - 1], m->a[s - 1], m->dpre[s])); }
      CHK(launch_dgrad(m, st, s, x, m->dpre[s], m->gha));
    }
    CHK(launch_agg(m, st, d, r, m->N, F, m->gha + F, 2 * F, m->gha, 2 * F, m->h[0], m->dpre[0], 1));
  }
  if (split) { CHK(fork()); CHK(wgrad_gnn(m, sw, 0, x, d.xe, nullptr, d.nbr, m->dpre[0])); }
  else if (merged) CHK(wgrad_all(m, st, x, d));             // every layer, one launch
  else CHK(wgrad_gnn_all(m, st, x, d));                     // all L+1 GNN stages, one launch
  if (two) {                      // join: st waits for the side stream
    hipEvent_t e = m->ev[evi++];
    HIPCHK(m, hipEventRecord(e, sw));
    HIPCHK(m, hipStreamWaitEvent(st, e, 0));
  }
  return V2X_OK;
}

// forward (+ backward) of the whole batch; everything is joined back into `st`
int run_step(v2x_model* m, hipStream_t st, const DevBatch& d, bool bwd, const float* y_dev, int n_global) {
  const Range all{0, d.B};
  CHK(run_forward(m, st, d, all, !(bwd && mlp_fused_training(m))));
  if (bwd) {
    // measured: with the current kernels the side-stream overlap of the Dense weight gradients no longer pays
    // (0.397 vs 0.390 ms/step), so one stream is the default; V2X_TWO_STREAMS=1 restores the fork/join
    const bool two = !m->prof && m->side && getenv("V2X_TWO_STREAMS") != nullptr;
    CHK(run_backward(m, st, two ? m->side : st, d, all, y_dev, n_global));
  }
  return V2X_OK;
}

int resolve_y(v2x_model* m, const float* y, int y_on_device, int R, hipStream_t st, const float** out) {
  if (!y) FAIL(m, V2X_EINVAL, "targets y are null");
  if (y_on_device) { *out = y; return V2X_OK; }
  CHK(ensure(m, m->st_y, (size_t)R * m->C * sizeof(float)));
  HIPCHK(m, hipMemcpyAsync(m->st_y.p, y, (size_t)R * m->C * sizeof(float), hipMemcpyHostToDevice, st));
  *out = (const float*)m->st_y.p;
  return V2X_OK;
}

int emit_loss(v2x_model* m, float* loss_out, int loss_on_device, hipStream_t st) {
  if (!loss_out) return V2X_OK;
  const int n = m->cfg.variable_graphs ? 1 : m->N;
  if (loss_on_device) {
    HIPCHK(m, hipMemcpyAsync(loss_out, m->loss_dev, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    HIPCHK(m, hipMemcpyAsync(loss_out, m->loss_dev, n * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(m, hipStreamSynchronize(st));
    CHK(check_flag(m));
  }
  return V2X_OK;
}

// Run `body` either eagerly or through a cached hipGraph keyed on the pointers / sizes it bakes in.
// Capture happens on a PRIVATE stream (m->cap), not on the caller's: a hipGraph does not remember the stream it was recorded on,
// and the caller's stream is the one the host framework's collectives synchronise with -- RCCL's watchdog thread polls events
// that were last recorded on it, and polling such an event while the stream is capturing is an error that takes the process down
// (hipErrorCapturedEvent: seen once in five runs of tools/dp_host_overhead.py, round 6).  `st` is the caller's local stream
// variable, taken by reference: the body's lambdas read it, so it names the capture stream while the body records.
template <typename Body>
int run_maybe_graph(v2x_model* m, hipStream_t& st, const GraphKey& key, Body body) {
  const bool want = m->cfg.use_graph && !m->prof && st != nullptr;
  if (!want) return body();
  auto it = m->graphs.find(key);
  if (it != m->graphs.end()) {
    size_t i = 0;                                   // what the captured host code had left in the layer descriptors
    for (auto* v : {&m->gnn, &m->dense})
      for (LayerDesc& ld : *v) ld.n_slabs = it->second.n_slabs[i++];
    m->frag_live = it->second.frag_live;            // ... and in the model (layout of the saved h_L / a_L / gha)
    HIPCHK(m, hipGraphLaunch(it->second.exec, st));
    return V2X_OK;
  }
  // make sure every lazily-sized buffer and function attribute exists before capture: run once eagerly
  // is not possible without side effects on optimizer state, so callers pre-size buffers (see below).
  hipGraph_t g = nullptr;
  hipStream_t const user = st, cap = m->cap ? m->cap : st;
  HIPCHK(m, hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  m->capturing = true;
  st = cap;
  const int r = body();
  st = user;
  m->capturing = false;
  hipError_t e = hipStreamEndCapture(cap, &g);
  if (r != V2X_OK) { if (g) hipGraphDestroy(g); return r; }
  if (e != hipSuccess) FAIL(m, V2X_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t ge = nullptr;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) FAIL(m, V2X_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  if (m->graphs.size() >= 16) drop_graphs(m);
  v2x_model::GraphEntry entry;
  entry.exec = ge;
  entry.frag_live = m->frag_live;
  for (auto* v : {&m->gnn, &m->dense})
    for (const LayerDesc& ld : *v) entry.n_slabs.push_back(ld.n_slabs);
  m->graphs[key] = entry;
  HIPCHK(m, hipGraphLaunch(ge, st));
  return V2X_OK;
}

GraphKey make_key(int kind, const DevBatch& d, const void* y, int n_global) {
  GraphKey k;
  memset(&k, 0, sizeof(k));
  k.kind = kind;
  k.ptrs[0] = d.xe; k.ptrs[1] = d.nbr; k.ptrs[2] = d.goff; k.ptrs[3] = d.rp; k.ptrs[4] = d.ci; k.ptrs[5] = y;
  k.sizes[0] = d.B; k.sizes[1] = d.R; k.sizes[2] = d.E; k.sizes[3] = d.max_nodes; k.sizes[4] = d.max_edges;
  k.sizes[5] = n_global;
  return k;
}

int max_slabs(const v2x_model* m, int n_idx, int n_slots) {
  int chunk, nc = 1;
  const int mr = m->L >= 1 ? merged_wg_rows(m, n_idx, n_slots) : 0;
  for (int rows : {env_int("V2X_WG_CHUNK_GNN", 1024), env_int("V2X_WG_CHUNK_DENSE", 1024), env_int("V2X_WG_CHUNK_EMBED", 1024), env_int("V2X_WG_CHUNK_D123", 512), 768, 896, mr > 0 ? mr : 1024})
    if (rows > 0) nc = std::max(nc, role_chunks(n_idx, n_slots, 1000, 1000, &chunk, rows));     // (a switch set to 0 = its default)
  if (is_wide(m)) nc = std::max(nc, wide_splits(n_idx, 1, n_slots));     // the fewest tiles (one) split most
  else nc = std::max(nc, mlp_wg_split(n_idx, n_slots).n_slabs);         // k_mlp_train_wg: one slab per workgroup and slot
  return nc + 1;
}

int presize(v2x_model* m, const DevBatch& d) {
  CHK(ensure_rows(m, d.R));
  if (need_adj_masks(m, d)) CHK(ensure(m, m->adj_mask, (size_t)2 * d.R * ((d.max_nodes + 31) / 32) * 4));
  if (ragged_fused_path(m, d)) CHK(ensure(m, m->plan_buf, (size_t)(ragged_wgs(m, d) + 2) * 4));
  if (fused_split(m, d) > 1) CHK(ensure_xchg(m, (d.B + FZ_TG - 1) / FZ_TG));
  const IdxMap x = idx_map(m, d, Range{0, d.B});
  CHK(ensure_slabs(m, max_slabs(m, x.n_idx, x.grid_y)));
  return V2X_OK;
}

// helpers for the per-kernel entry points: the whole batch as one range
IdxMap idx_map_rows(const v2x_model* m, int n_rows) {
  DevBatch d;
  memset(&d, 0, sizeof(d));
  d.R = n_rows; d.B = m->cfg.variable_graphs ? 1 : n_rows / m->N;
  return idx_map(m, d, Range{0, d.B});
}

int presize_rows(v2x_model* m, int n_rows) {
  CHK(ensure_rows(m, n_rows));
  const IdxMap x = idx_map_rows(m, n_rows);
  CHK(ensure_slabs(m, max_slabs(m, x.n_idx, x.grid_y)));
  return V2X_OK;
}

// Adam in the weight-gradient epilogues (single-GPU fit step of a wide model): needs the merged launch and ONE row split
// for every layer it covers (gnn[1..L], dense[0]) -- true as soon as K tiles x slots fill the chip (per-node weights)
bool wide_adam_fusable(const v2x_model* m, const DevBatch& d) {
  static const int on = env_int("V2X_WIDE_ADAM", 1), wide_merge = env_int("V2X_WIDE_MERGE", 1);
  if (!on || !wide_merge || !is_wide(m) || m->bucketed || m->prof_no_fuse || !m->adam_scal || m->L + 2 > WWM_ROLES) return false;
  if (getenv("V2X_TWO_STREAMS")) return false;
  const IdxMap x = idx_map(m, d, Range{0, d.B});
  const int kt = 2 * ((m->F + 127) / 128);                    // [h | agg] K tiles ([x | e] folded or one more: more tiles = fewer splits)
  return wide_splits(x.n_idx, kt, x.grid_y) == 1;
}

}  // namespace

// ======================================================================================= C ABI
extern "C" {

const char* v2x_version(void) { return "v2xgnn 0.1 (gfx950)"; }

const char* v2x_last_error(const v2x_model* m) { return m ? m->err.c_str() : g_err.c_str(); }

int v2x_create(const v2x_config* cfg, v2x_model** out) {
  v2x_model* nullm = nullptr;
  if (!cfg || !out) FAIL(nullm, V2X_EINVAL, "v2x_create: null argument");
  *out = nullptr;
  if (cfg->n_channels != 4) FAIL(nullm, V2X_EINVAL, "n_channels must be 4 in this build (got %d)", cfg->n_channels);
  if (cfg->feat_dim != 16 && cfg->feat_dim != 32 && cfg->feat_dim != 64 && cfg->feat_dim != 128 && cfg->feat_dim != 256)
    FAIL(nullm, V2X_EINVAL, "feat_dim must be 16, 32, 64, 128 or 256 in this build (got %d)", cfg->feat_dim);
  if (cfg->n_nodes < 1) FAIL(nullm, V2X_EINVAL, "n_nodes must be >= 1");
  if (cfg->n_mp_layers < 1 || cfg->n_mp_layers > 8) FAIL(nullm, V2X_EINVAL, "n_mp_layers must be in [1,8]");
  if (cfg->variable_graphs && !cfg->share_weights)
    FAIL(nullm, V2X_EINVAL, "variable_graphs requires share_weights (per-node weights need a fixed node count)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    FAIL(nullm, V2X_EHIP, "no HIP device available: the v2xgnn engine has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) FAIL(nullm, V2X_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
  if (hipSetDevice(cfg->device) != hipSuccess) FAIL(nullm, V2X_EHIP, "hipSetDevice(%d) failed", cfg->device);

  v2x_model* m = new v2x_model();
  m->cfg = *cfg;
  if (m->cfg.lr <= 0.f) m->cfg.lr = 1e-3f;
  if (m->cfg.beta1 <= 0.f) m->cfg.beta1 = 0.5f;      // BS_brain.py:212
  if (m->cfg.beta2 <= 0.f) m->cfg.beta2 = 0.999f;
  if (m->cfg.eps <= 0.f) m->cfg.eps = 1e-7f;         // K.epsilon()
  m->N = cfg->n_nodes; m->C = cfg->n_channels; m->F = cfg->feat_dim; m->L = cfg->n_mp_layers;
  m->S = cfg->share_weights ? 1 : cfg->n_nodes;
  m->Dn = 2 * m->C + 1; m->De = m->C;               // BS_brain.py:101-102 with node_info=3, edge_info=1
  build_layout(m);
  set_attrs(m->F);
  m->h.assign(m->L + 1, nullptr);
  m->a.assign(m->L + 1, nullptr);
  m->dpre.assign(m->L + 1, nullptr);
  auto fail = [&](const char* what) {
    g_err = std::string("v2x_create: ") + what + ": " + m->err;
    v2x_destroy(m);
    return V2X_EHIP;
  };
  const size_t pb = (size_t)m->P * sizeof(float);
  if (dev_alloc(m, &m->params, m->P) || dev_alloc(m, &m->grads, m->P) || dev_alloc(m, &m->mom, m->P) ||
      dev_alloc(m, &m->vel, m->P) || dev_alloc(m, &m->loss_dev, (size_t)m->N + 1) || dev_alloc(m, &m->zero_buf, 1024) ||
      dev_alloc(m, &m->loss_part, 128) || dev_alloc(m, &m->adam_scal, 16))
    return fail("allocation");
  // V2X_FUSED=0 (read when the model is created) keeps the layer-by-layer kernels: A/B measurements and the test that
  // the two paths agree bitwise
  m->compl_sums = env_int("V2X_FUSED_COMPL", 1) != 0;
  m->small_predict = env_int("V2X_SMALL_PREDICT", 1) != 0;
  m->split_env = env_int("V2X_FUSED_SPLIT", -1);
  m->ragged_fused = env_int("V2X_RAGGED_FUSED", 1) != 0;
  m->ragged_fused_bwd = env_int("V2X_RAGGED_FUSED_BWD", 1) != 0;
  m->ragged_packed = env_int("V2X_RAGGED_PACKED", 1) != 0;
  m->ragged_small_env = env_int("V2X_RAGGED_SMALL", 0) != 0;
  m->ragged_plan_fold = env_int("V2X_RAGGED_PLAN_FOLD", 1) != 0;
  if (m->small_predict && !m->cfg.variable_graphs && m->F <= 64 && m->L <= FZ_MAXL) {
    const size_t hb = (size_t)(m->L + 1) * SMALL_ROWS * m->F * sizeof(unsigned long long), sb = (size_t)2 * SMALL_ROWS * sizeof(unsigned);
    void *ph = nullptr, *ps = nullptr;
    if (hipMalloc(&ph, hb) != hipSuccess || hipMalloc(&ps, sb) != hipSuccess) return fail("allocation");
    m->small_h = static_cast<unsigned long long*>(ph);
    if (hipMemset(ph, 0, hb)) return fail("memset");             // tag 0 = never written
    m->small_sync = static_cast<unsigned long long*>(ps);
    void *hh = nullptr, *hdv = nullptr;
    if (env_int("V2X_SMALL_PINNED", 1) != 0 && hipHostMalloc(&hh, PIN_BYTES, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(&hdv, hh, 0) == hipSuccess) {
      m->pin_h = static_cast<char*>(hh); m->pin_d = static_cast<char*>(hdv);
    } else {
      (void)hipGetLastError();
      if (hh) hipHostFree(hh);
    }
    if (hipMemset(m->small_sync, 0, (size_t)2 * SMALL_ROWS * sizeof(unsigned))) return fail("memset");
  }
  // (ragged models with shared weights stream the same copy in their small-tile kernels, kernels_ragged_small.hpp)
  if (env_int("V2X_FUSED", 1) != 0 && (!m->cfg.variable_graphs || (m->S == 1 && m->ragged_small_env)) && m->F <= 64 && m->L <= FZ_MAXL) {
    const int FB = m->F / 16, KB = 2 * FB + 1;
    const size_t fwd0 = (size_t)FB * 256 + m->F, fwd = (size_t)KB * FB * 256 + m->F, bwd = (size_t)FB * 2 * FB * 256;
    if (dev_alloc(m, &m->pk_fwd, (size_t)m->S * fwd0 + (size_t)m->L * m->S * fwd) || dev_alloc(m, &m->pk_bwd, (size_t)m->L * m->S * bwd))
      return fail("allocation");
    m->pk_stale = true;             // first forward packs whatever the parameters are by then
    if (m->F == 64 && !m->cfg.variable_graphs && dev_alloc(m, &m->mlp_img, (size_t)m->S * mlp_image_floats<64>())) return fail("allocation");
  }
  if ((m->pk_fwd || m->cfg.variable_graphs) && env_int("V2X_FUSED_TS", 0)) {
    if (dev_alloc(m, &m->ts_buf, 4 * 8 * 64)) return fail("allocation");
    hipMemset(m->ts_buf, 0, 4 * 8 * 64 * 8);
  }
  if (hipMemset(m->zero_buf, 0, 4096) || hipMemset(m->loss_part, 0, 512) || hipMemset(m->params, 0, pb) || hipMemset(m->grads, 0, pb) || hipMemset(m->mom, 0, pb) || hipMemset(m->vel, 0, pb))
    return fail("memset");
  {
    FlagWord f;
    if (alloc_flag(&f) != V2X_OK) return fail("pinned flag word");
    m->flag_host = f.host; m->flag_dev = f.dev;
  }
  if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) return fail("side stream");
  if (hipStreamCreateWithFlags(&m->cap, hipStreamNonBlocking) != hipSuccess) return fail("capture stream");
  m->ev.resize(2 * m->L + 6);
  for (auto& e : m->ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail("event");
  *out = m;
  return V2X_OK;
}

void v2x_destroy(v2x_model* m) {
  if (!m) return;
  hipSetDevice(m->cfg.device);
  hipDeviceSynchronize();
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second.exec);
  for (auto& r : m->prof_recs) { hipEventDestroy(r.ev0); hipEventDestroy(r.ev1); }
  float* ptrs[] = {m->params, m->grads, m->mom, m->vel, m->z1, m->z2, m->z3, m->q, m->dq, m->dz1, m->dz2, m->dz3,
                   m->gha, m->rowloss, m->loss_dev, m->slab, m->zero_buf, m->loss_part, m->pk_fwd, m->pk_bwd, m->adam_scal, m->mlp_img};
  for (float* p : m->dpre) if (p) hipFree(p);
  if (m->gate_bits) hipFree(m->gate_bits);
  if (m->nbmask) hipFree(m->nbmask);
  for (auto& e : m->ev) if (e) hipEventDestroy(e);
  if (m->side) hipStreamDestroy(m->side);
  if (m->cap) hipStreamDestroy(m->cap);
  for (float* p : ptrs) if (p) hipFree(p);
  for (float* p : m->h) if (p) hipFree(p);
  for (float* p : m->a) if (p) hipFree(p);
  DevBuf* bufs[] = {&m->st_xe, &m->st_nbr, &m->st_goff, &m->st_rp, &m->st_ci, &m->st_y, &m->st_q, &m->adj_mask, &m->plan_buf};
  for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
  if (m->flag_host) hipHostFree(m->flag_host);
  if (m->ts_buf) hipFree(m->ts_buf);
  if (m->small_h) hipFree(m->small_h);
  if (m->xchg_sync) {                  // the process-wide epoch base moves past everything this buffer's address has seen
    unsigned long long seen = 0;
    if (xchg_max_count(m, &seen) == V2X_OK) {
      std::lock_guard<std::mutex> lock(g_xchg_epoch_mutex);
      g_xchg_epoch_next = std::max(g_xchg_epoch_next, seen + 1);
    }
  }
  if (m->xchg_buf) hipFree(m->xchg_buf);
  if (m->xchg_sync) hipFree(m->xchg_sync);
  if (m->small_sync) hipFree(m->small_sync);
  if (m->pin_h) hipHostFree(m->pin_h);
  delete m;
}

int64_t v2x_param_count(const v2x_model* m) { return m ? m->P : 0; }
float* v2x_param_ptr(v2x_model* m) {
  if (m) {
    // the caller may write the parameters behind the library's back from now on: every fused forward re-packs its
    // fragment-major copy first -- a launch that graphs captured BEFORE this call do not contain, so they are dropped
    if (!m->raw_params) drop_graphs(m);
    m->pk_stale = m->raw_params = true;
  }
  return m ? m->params : nullptr;
}
float* v2x_grad_ptr(v2x_model* m) { return m ? m->grads : nullptr; }

int v2x_get_weights(v2x_model* m, float* host_out, void* stream) {
  if (!m || !host_out) FAIL(m, V2X_EINVAL, "get_weights: null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(m, hipMemcpyAsync(host_out, m->params, (size_t)m->P * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(m, hipStreamSynchronize(st));
  return V2X_OK;
}

int v2x_set_weights(v2x_model* m, const float* host_in, void* stream) {
  if (!m || !host_in) FAIL(m, V2X_EINVAL, "set_weights: null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(m, hipMemcpyAsync(m->params, host_in, (size_t)m->P * 4, hipMemcpyHostToDevice, st));
  if (m->pk_fwd) { CHK(launch_pack(m, st)); m->pk_stale = m->raw_params; }
  HIPCHK(m, hipStreamSynchronize(st));
  return V2X_OK;
}

int v2x_copy_weights(v2x_model* dst, const v2x_model* src, void* stream) {
  if (!dst || !src) FAIL(dst, V2X_EINVAL, "copy_weights: null argument");
  if (dst->P != src->P || dst->F != src->F || dst->N != src->N || dst->S != src->S || dst->L != src->L)
    FAIL(dst, V2X_EINVAL, "copy_weights: models have different shapes");
  HIPCHK(dst, hipMemcpyAsync(dst->params, src->params, (size_t)dst->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (dst->pk_fwd) { CHK(launch_pack(dst, (hipStream_t)stream)); dst->pk_stale = dst->raw_params; }
  return V2X_OK;
}

int v2x_get_optimizer_state(v2x_model* m, float* host_m, float* host_v, int64_t* iterations, void* stream) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  hipStream_t st = (hipStream_t)stream;
  if (host_m) HIPCHK(m, hipMemcpyAsync(host_m, m->mom, (size_t)m->P * 4, hipMemcpyDeviceToHost, st));
  if (host_v) HIPCHK(m, hipMemcpyAsync(host_v, m->vel, (size_t)m->P * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(m, hipStreamSynchronize(st));
  if (iterations) *iterations = m->iterations;
  return V2X_OK;
}

int v2x_set_optimizer_state(v2x_model* m, const float* host_m, const float* host_v, int64_t iterations, void* stream) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  hipStream_t st = (hipStream_t)stream;
  if (host_m) HIPCHK(m, hipMemcpyAsync(m->mom, host_m, (size_t)m->P * 4, hipMemcpyHostToDevice, st));
  if (host_v) HIPCHK(m, hipMemcpyAsync(m->vel, host_v, (size_t)m->P * 4, hipMemcpyHostToDevice, st));
  HIPCHK(m, hipStreamSynchronize(st));
  m->iterations = iterations;
  return V2X_OK;
}

int v2x_forward(v2x_model* m, const v2x_batch* b, float* q_out, int q_on_device, void* stream) {
  if (!m || !q_out) FAIL(m, V2X_EINVAL, "forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(m, hipSetDevice(m->cfg.device));
  // host batch in, host q out, a few graphs (BS.predict in the rollout loop): the kernel reads the batch from, and writes q
  // to, a pinned device-mapped window -- one launch and one stream synchronisation, no copy launches at all
  if (b && !b->on_device && !q_on_device && m->pin_h && b->xe && b->row_ptr && b->n_rows > 0 && b->n_rows <= SMALL_ROWS &&
      b->n_edges >= 0 && (size_t)b->n_edges * 4 <= PIN_Q - PIN_CI && (b->n_edges == 0 || b->col_idx)) {
    DevBatch hd;
    memset(&hd, 0, sizeof(hd));
    hd.B = b->n_graphs; hd.R = b->n_rows; hd.E = b->n_edges; hd.max_nodes = b->max_nodes; hd.max_edges = b->max_edges;
    hd.goff = b->graph_off; hd.nbr = b->nbr_init;
    if (hd.B > 0 && !m->cfg.variable_graphs && hd.R == hd.B * m->N && small_path(m, hd)) {
      CHK(validate_host_batch(m, b, m->N));
      memcpy(m->pin_h + PIN_XE, b->xe, (size_t)hd.R * XE * sizeof(float));
      memcpy(m->pin_h + PIN_RP, b->row_ptr, (size_t)(hd.R + 1) * 4);
      if (hd.E > 0) memcpy(m->pin_h + PIN_CI, b->col_idx, (size_t)hd.E * 4);
      hd.xe = reinterpret_cast<const float*>(m->pin_d + PIN_XE);
      hd.rp = reinterpret_cast<const int32_t*>(m->pin_d + PIN_RP);
      hd.ci = reinterpret_cast<const int32_t*>(m->pin_d + PIN_CI);
      float* qd = reinterpret_cast<float*>(m->pin_d + PIN_Q);
      CHK(run_maybe_graph(m, st, make_key(8, hd, qd, 0), [&]() { return launch_small_forward(m, st, hd, qd); }));
      m->have_fwd = false;
      HIPCHK(m, hipStreamSynchronize(st));
      CHK(check_flag(m));
      memcpy(q_out, m->pin_h + PIN_Q, (size_t)hd.R * m->C * sizeof(float));
      return V2X_OK;
    }
  }
  DevBatch d;
  CHK(resolve_batch(m, b, &d, st));
  CHK(presize(m, d));
  // a few graphs (the rollout predict): the whole network in one launch, nothing saved for a backward pass
  const bool small = small_path(m, d);
  if (small) CHK(run_maybe_graph(m, st, make_key(7, d, nullptr, 0), [&]() { return launch_small_forward(m, st, d); }));
  else CHK(run_maybe_graph(m, st, make_key(1, d, nullptr, 0), [&]() { return run_step(m, st, d, false, nullptr, 0); }));
  m->have_fwd = !small;
  const size_t qb = (size_t)d.R * m->C * sizeof(float);
  if (q_on_device) {
    HIPCHK(m, hipMemcpyAsync(q_out, m->q, qb, hipMemcpyDeviceToDevice, st));
  } else {
    HIPCHK(m, hipMemcpyAsync(q_out, m->q, qb, hipMemcpyDeviceToHost, st));
    HIPCHK(m, hipStreamSynchronize(st));
    CHK(check_flag(m));
  }
  return V2X_OK;
}

static int fwd_bwd(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device, int32_t n_global,
                   float* loss_out, int loss_on_device, void* stream, bool with_adam) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(m, hipSetDevice(m->cfg.device));
  DevBatch d;
  CHK(resolve_batch(m, b, &d, st));
  if (n_global <= 0) n_global = m->cfg.variable_graphs ? d.R : d.B;
  const float* yd;
  CHK(resolve_y(m, y, y_on_device, d.R, st, &yd));
  CHK(presize(m, d));
  if (with_adam) {
    // lr_t depends on the iteration count, so the Adam node stays outside the replayed graph.  Wide path: Adam rides on the
    // weight-gradient launch for the layers that launch writes in place; this step's lr_t reaches the (replayed) kernels
    // through device memory, written by a one-thread launch whose ARGUMENTS carry the values (copied when the launch is
    // enqueued: no asynchronous read of host stack memory, no host / stream synchronisation -- ADVICE r04)
    const bool fuse = wide_adam_fusable(m, d);
    if (fuse) {
      hipLaunchKernelGGL(k_set4, dim3(1), dim3(64), 0, st, m->adam_scal, adam_lr_t(m, m->iterations + 1), m->cfg.beta1, m->cfg.beta2, m->cfg.eps);
      CHK(launch_check(m, "k_set4"));
    }
    m->fuse_adam_now = fuse;
    const int rc = run_maybe_graph(m, st, make_key(fuse ? 9 : 2, d, yd, n_global), [&]() { return run_step(m, st, d, true, yd, n_global); });
    m->fuse_adam_now = false;
    CHK(rc);
    CHK(launch_reduce_adam(m, st, 1, true, nullptr, loss_job(m, d, n_global), -1, fuse));
  } else {
    CHK(run_maybe_graph(m, st, make_key(3, d, yd, n_global), [&]() {
      CHK(run_step(m, st, d, true, yd, n_global));
      return launch_reduce_adam(m, st, 1, false, nullptr, loss_job(m, d, n_global));
    }));
  }
  m->have_fwd = true;
  return emit_loss(m, loss_out, loss_on_device, st);
}

// Data-parallel step in two phases (dp.py): phase 0 = forward, decision-MLP forward + Huber + backward, the Dense layers'
// weight gradients reduced into their bucket of the flat gradient; phase 1 = graph-layer backward, their weight gradients,
// the per-output losses.  After phase 0 the Dense bucket is final, so the host can start its all-reduce (RCCL runs on its
// own stream) while phase 1 computes.
int v2x_forward_backward_phase(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device, int32_t n_global, int phase,
                               float* loss_out, int loss_on_device, void* stream) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  if (phase < 0 || phase >= n_phase_buckets(m)) FAIL(m, V2X_EINVAL, "forward_backward_phase: phase must be in [0, %d)", n_phase_buckets(m));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(m, hipSetDevice(m->cfg.device));
  DevBatch d;
  CHK(resolve_batch(m, b, &d, st));
  if (n_global <= 0) n_global = m->cfg.variable_graphs ? d.R : d.B;
  const float* yd;
  CHK(resolve_y(m, y, y_on_device, d.R, st, &yd));
  CHK(presize(m, d));
  const Range all{0, d.B};
  const IdxMap x = idx_map(m, d, all);
  const int F = m->F, L = m->L;
  if (is_wide(m)) {
    // one weight-gradient launch per layer: phase 0 = forward, decision MLP backward, the Dense layers' gradients; phase k =
    // the data gradient handed down by stage s + 1 (not before: the all-reduce of the bucket a phase completes starts when the
    // phase's LAST launch is done), the transposed aggregation and the weight gradient of stage s = L + 1 - k
    if (phase > 0 && !m->have_fwd) FAIL(m, V2X_ESTATE, "forward_backward_phase: phase %d before phase 0", phase);
    m->bucketed = true;
    const int rc = run_maybe_graph(m, st, make_key(20 + phase, d, yd, n_global), [&]() -> int {
      int64_t rb, re;
      phase_bucket_range(m, phase, &rb, &re);
      if (phase == 0) {
        CHK(run_forward(m, st, d, all, true));
        MlpArgs a;
        mlp_args(m, a, x, d.xe, m->h[L], m->a[L]);
        a.y = yd;
        a.inv_denom = 1.0f / loss_denominator(m, n_global);
        CHK(launch_mlp(m, st, a, true));
        CHK(wgrad_mlp(m, st, x, d.xe, m->h[L], m->a[L]));
        return launch_reduce_adam(m, st, 1, false, nullptr, LossJob{0, 0, 0, 0.f, 0}, -1, false, rb, re);
      }
      const int s = L + 1 - phase;
      if (s < L) CHK(launch_dgrad(m, st, s + 1, x, m->dpre[s + 1], m->gha));
      CHK(launch_agg(m, st, d, all, m->N, F, m->gha + F, 2 * F, m->gha, 2 * F, (s < L) ? m->h[s] : nullptr, m->dpre[s], 1));
      CHK(wide_wgrad_gnn(m, st, s, x, d.xe, s ? m->h[s - 1] : nullptr, s ? m->a[s - 1] : d.nbr, m->dpre[s]));
      const bool last = s == 0;
      if (m->gnn[s].n_slabs > 0 || last)
        return launch_reduce_adam(m, st, 1, false, nullptr, last ? loss_job(m, d, n_global) : LossJob{0, 0, 0, 0.f, 0}, -1, false, rb, re);
      return V2X_OK;
    });
    m->bucketed = false;
    CHK(rc);
    if (phase == 0) m->have_fwd = true;
    return phase == n_phase_buckets(m) - 1 ? emit_loss(m, loss_out, loss_on_device, st) : V2X_OK;
  }
  if (phase == 0) {
    CHK(run_maybe_graph(m, st, make_key(5, d, yd, n_global), [&]() -> int {
      CHK(run_forward(m, st, d, all, !mlp_fused_training(m)));
      MlpArgs a;
      mlp_args(m, a, x, d.xe, m->h[L], m->a[L]);
      a.y = yd;
      a.inv_denom = 1.0f / loss_denominator(m, n_global);
      a.frag_groups = m->frag_live ? d.B / FZ_TG : 0;
      if (mlp_wg_path(m)) {
        CHK(launch_mlp_train_wg(m, st, a));
      } else {
        if (mlp_fused_training(m)) CHK(launch_mlp_train(m, st, a));
        else CHK(launch_mlp(m, st, a, true));
        CHK(wgrad_mlp(m, st, x, d.xe, m->h[L], m->a[L]));
      }
      return launch_reduce_adam(m, st, 1, false, nullptr, LossJob{0, 0, 0, 0.f, 0}, 0);
    }));
    m->have_fwd = true;
    return V2X_OK;
  }
  if (!m->have_fwd) FAIL(m, V2X_ESTATE, "forward_backward_phase: phase 1 before phase 0");
  CHK(run_maybe_graph(m, st, make_key(6, d, yd, n_global), [&]() -> int {
    if (fused_path(m, d)) {
      CHK(launch_fused_bwd(m, st, d));
    } else {
      for (int s = L; s >= 1; --s) {
        CHK(launch_agg(m, st, d, all, m->N, F, m->gha + F, 2 * F, m->gha, 2 * F, s < L ? m->h[s] : nullptr, m->dpre[s], 1));
        CHK(launch_dgrad(m, st, s, x, m->dpre[s], m->gha));
      }
      CHK(launch_agg(m, st, d, all, m->N, F, m->gha + F, 2 * F, m->gha, 2 * F, m->h[0], m->dpre[0], 1));
    }
    CHK(wgrad_gnn_all(m, st, x, d));
    return launch_reduce_adam(m, st, 1, false, nullptr, loss_job(m, d, n_global), 1);
  }));
  return emit_loss(m, loss_out, loss_on_device, st);
}

int v2x_grad_bucket_count(const v2x_model* m) { return m ? n_phase_buckets(m) : 0; }

int64_t v2x_grad_bucket(const v2x_model* m, int bucket, int64_t* offset) {
  if (!m || bucket < 0 || bucket >= n_phase_buckets(m)) return 0;
  int64_t b, e;
  phase_bucket_range(m, bucket, &b, &e);
  if (offset) *offset = b;
  return e - b;
}

int v2x_apply_gradients_range(v2x_model* m, int64_t offset, int64_t count, int advance_iteration, void* stream) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  if (offset < 0 || count <= 0 || offset + count > m->P || (offset & 3) || (count & 3))
    FAIL(m, V2X_EINVAL, "apply_gradients_range: [%lld, +%lld) must lie inside the %lld parameters and be float4-aligned",
         (long long)offset, (long long)count, (long long)m->P);
  HIPCHK(m, hipSetDevice(m->cfg.device));
  return launch_reduce_adam(m, (hipStream_t)stream, 0, true, nullptr, LossJob{0, 0, 0, 0.f, 0}, -1, false, offset, offset + count,
                            advance_iteration != 0);
}

int v2x_forward_call(void* closure) {
  v2x_forward_closure* c = (v2x_forward_closure*)closure;
  v2x_model* nullm = nullptr;
  if (!c) FAIL(nullm, V2X_EINVAL, "forward_call: null closure");
  return v2x_forward(c->m, &c->b, c->q_out, c->q_on_device, c->stream);
}

int v2x_train_step(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device, int32_t n_graphs_global,
                   float* loss_out, int loss_on_device, void* stream) {
  return fwd_bwd(m, b, y, y_on_device, n_graphs_global, loss_out, loss_on_device, stream, true);
}

int v2x_forward_backward(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device, int32_t n_graphs_global,
                         float* loss_out, int loss_on_device, void* stream) {
  return fwd_bwd(m, b, y, y_on_device, n_graphs_global, loss_out, loss_on_device, stream, false);
}

// One DQN replay step (Agent.replay, BS_brain.py:555-748) in a single call: target forward on s', online forward on s
// (its activations are kept), y = q with the taken action's entry replaced by r + gamma * max q', then backward +
// Adam on the online network WITHOUT a second forward of the graph layers (predict + fit would run them twice).
int v2x_dqn_step(v2x_model* online, v2x_model* target, const v2x_batch* s, const v2x_batch* s_next, const int32_t* action,
                 const double* reward, double gamma, int32_t n_graphs_global, float* y_out, float* loss_out,
                 int loss_on_device, void* stream) {
  v2x_model* m = online;
  if (!online || !target || !action || !reward) FAIL(m, V2X_EINVAL, "dqn_step: null argument");
  if (online == target) FAIL(m, V2X_EINVAL, "dqn_step: online and target must be different models");
  if (online->cfg.variable_graphs || target->N != online->N || target->F != online->F || target->L != online->L ||
      target->S != online->S || target->cfg.device != online->cfg.device)
    FAIL(m, V2X_EINVAL, "dqn_step: fixed-size graphs and two models of the same shape on the same device are required");
  if (!s || !s_next || !s->on_device || !s_next->on_device) FAIL(m, V2X_EINVAL, "dqn_step takes device-resident batches");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(m, hipSetDevice(online->cfg.device));
  DevBatch ds, dn;
  CHK(resolve_batch(online, s, &ds, st));
  if (int rc = resolve_batch(target, s_next, &dn, st)) { online->err = target->err; return rc; }
  if (dn.R != ds.R) FAIL(m, V2X_EINVAL, "dqn_step: s and s' have different sizes");
  if (n_graphs_global <= 0) n_graphs_global = ds.B;
  CHK(presize(online, ds));
  if (int rc = presize(target, dn)) { online->err = target->err; return rc; }
  CHK(ensure(online, online->st_y, (size_t)ds.R * online->C * sizeof(float)));
  float* y = y_out ? y_out : (float*)online->st_y.p;
  const Range all{0, ds.B};
  // Everything up to the slab reduction is one replayable hipGraph (the Adam launch stays outside: lr_t changes per
  // step).  The key holds every pointer the launches bake in -- both batches, action / reward / target buffers and the
  // TARGET model's workspace, whose re-allocation must not leave a stale graph in the online model's cache.
  GraphKey key = make_key(4, ds, y, n_graphs_global);
  key.ptrs[6] = dn.xe; key.ptrs[7] = dn.rp; key.ptrs[8] = dn.ci; key.ptrs[9] = action; key.ptrs[10] = reward;
  key.ptrs[11] = target->q;
  key.gen = target->ws_gen;                              // the graph bakes in EVERY workspace pointer of the target (h, a, z,
  key.scalar = gamma;                                    // masks, ...), not q only
  // Narrow models (k_mlp_train_wg): the online network's decision MLP runs ONCE, inside the training launch, which forms the
  // targets from its own forward output (MlpArgs::tq) -- as Keras' fit re-computes the prediction it was handed as target, so
  // that the untouched entries carry exactly zero error (BS_brain.py:664-692, :728); the separate MLP forward of the online
  // network, and the target kernel's pass over q, are gone (38 us of a 520-us step at 20 links, batch 4096).  V2X_DQN_FUSED_TARGETS=0:
  // the three-launch form (forward, k_dqn_targets, training launch on y).
  static const int fuse_env = env_int("V2X_DQN_FUSED_TARGETS", 1);
  const bool fuse_targets = fuse_env && mlp_wg_path(online) && !is_wide(online) && online->C == 4;
  CHK(run_maybe_graph(online, st, key, [&]() -> int {
    target->capturing = online->capturing;
    int rc = run_forward(target, st, dn, all, true);
    target->capturing = false;
    if (rc) { online->err = target->err; return rc; }
    if (!fuse_targets) {
      CHK(run_forward(online, st, ds, all, true));
      hipLaunchKernelGGL(k_dqn_targets, dim3((ds.R + 255) / 256), dim3(256), 0, st, online->q, target->q, action, reward, gamma,
                         ds.R, online->N, online->C, y);
      return run_backward(online, st, st, ds, all, y, n_graphs_global);
    }
    CHK(run_forward(online, st, ds, all, false));                          // graph layers only
    float* tq = online->dq;                                                // (a workspace this path does not use otherwise: [R][4])
    hipLaunchKernelGGL(k_dqn_tq, dim3((ds.R + 255) / 256), dim3(256), 0, st, target->q, action, reward, gamma, ds.R, online->N, online->C, tq);
    online->dqn_tq = tq; online->dqn_y = y;
    rc = run_backward(online, st, st, ds, all, y, n_graphs_global);
    online->dqn_tq = nullptr; online->dqn_y = nullptr;
    return rc;
  }));
  CHK(launch_reduce_adam(online, st, 1, true, nullptr, loss_job(online, ds, n_graphs_global)));
  online->have_fwd = target->have_fwd = true;
  return emit_loss(online, loss_out, loss_on_device, st);
}

int v2x_apply_gradients(v2x_model* m, void* stream) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  HIPCHK(m, hipSetDevice(m->cfg.device));
  return launch_reduce_adam(m, (hipStream_t)stream, 0, true, nullptr);
}

// ---------------------------------------------------------------------------- per-kernel entry points
static int batch_dev_only(const v2x_batch* b, DevBatch* d) {
  v2x_model* nullm = nullptr;
  if (!b || !b->on_device) FAIL(nullm, V2X_EINVAL, "per-kernel entry points take device-resident batches");
  if (!b->row_ptr || b->n_graphs <= 0 || b->n_rows <= 0 || b->max_nodes <= 0) FAIL(nullm, V2X_EINVAL, "batch: bad sizes");
  d->B = b->n_graphs; d->R = b->n_rows; d->E = b->n_edges; d->max_nodes = b->max_nodes; d->max_edges = b->max_edges;
  d->xe = b->xe; d->nbr = b->nbr_init; d->goff = b->graph_off; d->rp = b->row_ptr; d->ci = b->col_idx;
  return V2X_OK;
}

int v2x_agg_fwd(const v2x_batch* b, int32_t n_nodes, int32_t feat_dim, const float* h, float* out, void* stream) {
  DevBatch d;
  CHK(batch_dev_only(b, &d));
  return launch_agg(nullptr, (hipStream_t)stream, d, Range{0, d.B}, n_nodes, feat_dim, h, feat_dim, nullptr, 0, nullptr, out, 0);
}

int v2x_agg_bwd(const v2x_batch* b, int32_t n_nodes, int32_t feat_dim, const float* g, float* out, void* stream) {
  DevBatch d;
  CHK(batch_dev_only(b, &d));
  return launch_agg(nullptr, (hipStream_t)stream, d, Range{0, d.B}, n_nodes, feat_dim, g, feat_dim, nullptr, 0, nullptr, out, 1);
}

int v2x_node_update_fwd(v2x_model* m, int32_t stage, int32_t n_rows, const float* xe, const float* h_prev,
                        const float* agg_prev, float* out, void* stream) {
  if (!m || !xe || !out || stage < 0 || stage > m->L || n_rows <= 0) FAIL(m, V2X_EINVAL, "node_update_fwd: bad argument");
  if (m->S > 1 && n_rows % m->N) FAIL(m, V2X_EINVAL, "node_update_fwd: n_rows not a multiple of n_nodes");
  return launch_node_fwd(m, (hipStream_t)stream, stage, idx_map_rows(m, n_rows), xe, h_prev, agg_prev, out);
}

static int copy_cols(v2x_model* m, float* dst, int dst_w, const float* src, int src_w, int col, int rows, hipStream_t st) {
  HIPCHK(m, hipMemcpy2DAsync(dst, (size_t)dst_w * 4, src + col, (size_t)src_w * 4, (size_t)dst_w * 4, rows,
                             hipMemcpyDeviceToDevice, st));
  return V2X_OK;
}

int v2x_node_update_bwd(v2x_model* m, int32_t stage, int32_t n_rows, const float* xe, const float* h_prev,
                        const float* agg_prev, const float* dpre, float* dh_prev, float* dagg_prev, float* grad_out,
                        void* stream) {
  if (!m || !xe || !dpre || stage < 0 || stage > m->L || n_rows <= 0) FAIL(m, V2X_EINVAL, "node_update_bwd: bad argument");
  if (m->S > 1 && n_rows % m->N) FAIL(m, V2X_EINVAL, "node_update_bwd: n_rows not a multiple of n_nodes");
  hipStream_t st = (hipStream_t)stream;
  CHK(presize_rows(m, n_rows));
  const IdxMap x = idx_map_rows(m, n_rows);
  CHK(wgrad_gnn(m, st, stage, x, xe, h_prev, agg_prev, dpre));
  if (grad_out) CHK(launch_reduce_adam(m, st, 1, false, grad_out));
  if (stage > 0 && (dh_prev || dagg_prev)) {
    CHK(launch_dgrad(m, st, stage, x, dpre, m->gha));
    if (dh_prev) CHK(copy_cols(m, dh_prev, m->F, m->gha, 2 * m->F, 0, n_rows, st));
    if (dagg_prev) CHK(copy_cols(m, dagg_prev, m->F, m->gha, 2 * m->F, m->F, n_rows, st));
  }
  return V2X_OK;
}

int v2x_mlp_fwd(v2x_model* m, int32_t n_rows, const float* xe, const float* h, const float* agg, float* q_out, void* stream) {
  if (!m || !xe || !h || !agg || !q_out || n_rows <= 0) FAIL(m, V2X_EINVAL, "mlp_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  CHK(ensure_rows(m, n_rows));
  MlpArgs a;
  mlp_args(m, a, idx_map_rows(m, n_rows), xe, h, agg);
  CHK(launch_mlp(m, st, a, false));
  HIPCHK(m, hipMemcpyAsync(q_out, m->q, (size_t)n_rows * m->C * 4, hipMemcpyDeviceToDevice, st));
  return V2X_OK;
}

int v2x_mlp_huber_bwd(v2x_model* m, int32_t n_rows, int32_t n_global, const float* xe, const float* h, const float* agg,
                      const float* y, float* dh, float* dagg, float* grad_out, float* loss_out, void* stream) {
  if (!m || !xe || !h || !agg || !y || n_rows <= 0) FAIL(m, V2X_EINVAL, "mlp_huber_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  CHK(presize_rows(m, n_rows));
  const IdxMap x = idx_map_rows(m, n_rows);
  if (n_global <= 0) n_global = m->cfg.variable_graphs ? n_rows : n_rows / m->N;
  MlpArgs a;
  mlp_args(m, a, x, xe, h, agg);
  a.y = y;
  a.inv_denom = 1.0f / loss_denominator(m, n_global);
  if (mlp_wg_path(m)) {
    CHK(launch_mlp_train_wg(m, st, a));
  } else {
    if (mlp_fused_training(m)) {
      CHK(launch_mlp_train(m, st, a));
    } else {
      CHK(launch_mlp(m, st, a, false));
      CHK(launch_mlp(m, st, a, true));
    }
    CHK(wgrad_mlp(m, st, x, xe, h, agg));
  }
  if (grad_out) CHK(launch_reduce_adam(m, st, 1, false, grad_out));
  if (dh) CHK(copy_cols(m, dh, m->F, m->gha, 2 * m->F, 0, n_rows, st));
  if (dagg) CHK(copy_cols(m, dagg, m->F, m->gha, 2 * m->F, m->F, n_rows, st));
  if (loss_out) {
    if (m->cfg.variable_graphs)
      hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, st, m->rowloss, m->loss_dev, n_rows, 1, (int64_t)0, a.inv_denom);
    else
      hipLaunchKernelGGL(k_loss_reduce, dim3(m->N), dim3(256), 0, st, m->rowloss, m->loss_dev, n_rows / m->N,
                         m->S == 1 ? m->N : 1, m->S == 1 ? (int64_t)1 : srow_stride(m), a.inv_denom);
    HIPCHK(m, hipMemcpyAsync(loss_out, m->loss_dev, (m->cfg.variable_graphs ? 1 : m->N) * sizeof(float),
                             hipMemcpyDeviceToDevice, st));
  }
  return V2X_OK;
}

int v2x_adam_step(float* param, const float* grad, float* mom, float* vel, int64_t n, int64_t iteration, float lr,
                  float beta1, float beta2, float eps, void* stream) {
  v2x_model* nullm = nullptr;
  if (!param || !grad || !mom || !vel || n <= 0 || iteration < 1) FAIL(nullm, V2X_EINVAL, "adam_step: bad argument");
  const double t = (double)iteration;
  const float lr_t = (float)(lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_adam_scalar, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, mom, vel, n, lr_t,
                     beta1, beta2, eps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) FAIL(nullm, V2X_EHIP, "adam launch failed: %s", hipGetErrorString(e));
  return V2X_OK;
}

// ---------------------------------------------------------------------------- DQN replay glue
int v2x_device_addressable(const void* p) {
  v2x_model* nullm = nullptr;
  if (!p) FAIL(nullm, V2X_EINVAL, "device_addressable: null pointer");
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }     // unknown to the runtime: pageable host memory
  if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) return 1;
  if (at.type == hipMemoryTypeHost) return at.devicePointer == p ? 1 : 0;
  return 0;
}

int v2x_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_idx, int64_t row_bytes, void* stream) {
  v2x_model* nullm = nullptr;
  if (!src || !idx || !dst || n_idx <= 0 || row_bytes <= 0 || (row_bytes & 3))
    FAIL(nullm, V2X_EINVAL, "gather_rows: null pointer, empty selection or row_bytes not a multiple of 4");
  const int64_t words = row_bytes / 4;
  const bool vec = (row_bytes & 15) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
  const int64_t total = n_idx * (vec ? words / 4 : words);
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  if (vec)
    hipLaunchKernelGGL(k_gather_rows<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src, idx,
                       (uint32_t*)dst, n_idx, words);
  else
    hipLaunchKernelGGL(k_gather_rows<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src, idx,
                       (uint32_t*)dst, n_idx, words);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) FAIL(nullm, V2X_EHIP, "gather_rows launch failed: %s", hipGetErrorString(e));
  return V2X_OK;
}

int v2x_gather_rows_multi(int32_t n_jobs, const void* const* src, void* const* dst, const int64_t* row_bytes, const int32_t* idx,
                          int64_t n_idx, void* stream) {
  v2x_model* nullm = nullptr;
  if (n_jobs < 1 || n_jobs > GATHER_MAX_JOBS || !src || !dst || !row_bytes || !idx || n_idx <= 0)
    FAIL(nullm, V2X_EINVAL, "gather_rows_multi: 1..%d jobs, non-null arrays, a non-empty selection", GATHER_MAX_JOBS);
  GatherJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  int64_t most = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!src[j] || !dst[j] || row_bytes[j] <= 0 || (row_bytes[j] & 3)) FAIL(nullm, V2X_EINVAL, "gather_rows_multi: job %d: null pointer or row_bytes not a multiple of 4", j);
    jobs.src[j] = (const uint32_t*)src[j]; jobs.dst[j] = (uint32_t*)dst[j]; jobs.words[j] = row_bytes[j] / 4;
    jobs.vec[j] = ((row_bytes[j] & 15) == 0 && ((uintptr_t)src[j] & 15) == 0 && ((uintptr_t)dst[j] & 15) == 0) ? 1 : 0;
    most = std::max(most, n_idx * (jobs.vec[j] ? jobs.words[j] / 4 : jobs.words[j]));
  }
  const int blocks = (int)std::min<int64_t>((most + 255) / 256, 2048);
  hipLaunchKernelGGL(k_gather_rows_multi, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs, idx, n_idx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) FAIL(nullm, V2X_EHIP, "gather_rows_multi launch failed: %s", hipGetErrorString(e));
  return V2X_OK;
}

int v2x_q_stats(const float* y, int32_t n_graphs, int32_t n_nodes, int32_t n_channels, double* out, void* stream) {
  v2x_model* nullm = nullptr;
  if (!y || !out || n_graphs <= 0 || n_nodes <= 0 || n_channels <= 0) FAIL(nullm, V2X_EINVAL, "q_stats: bad argument");
  // scratch of the two-level sum: per (device, stream) -- launches on one stream are ordered, two streams must not share the
  // arrival counters --, grown on demand, zeroed when allocated (the kernel re-arms its counters)
  static std::mutex mu;
  static std::map<std::pair<int, void*>, std::pair<void*, int>> scratch;          // -> (buffer, links it holds)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) FAIL(nullm, V2X_EHIP, "q_stats: no device");
  double* part = nullptr;
  unsigned* cnt = nullptr;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto& e = scratch[std::make_pair(dev, stream)];
    if (e.second < n_nodes) {
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) FAIL(nullm, V2X_EHIP, "q_stats: synchronise failed");
      if (e.first) (void)hipFree(e.first);
      const int cap = std::max(n_nodes, 64);
      const size_t bytes = (size_t)cap * QS_PARTS * 2 * sizeof(double) + (size_t)cap * sizeof(unsigned);
      if (hipMalloc(&e.first, bytes) != hipSuccess) { e.first = nullptr; e.second = 0; FAIL(nullm, V2X_ENOMEM, "q_stats: scratch"); }
      if (hipMemset(e.first, 0, bytes) != hipSuccess) FAIL(nullm, V2X_EHIP, "q_stats: memset failed");
      e.second = cap;
    }
    part = (double*)e.first;
    cnt = (unsigned*)(part + (size_t)e.second * QS_PARTS * 2);
  }
  hipLaunchKernelGGL(k_q_stats, dim3(n_nodes, QS_PARTS), dim3(256), 0, (hipStream_t)stream, y, n_graphs, n_nodes, n_channels, out, part, cnt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) FAIL(nullm, V2X_EHIP, "q_stats launch failed: %s", hipGetErrorString(e));
  return V2X_OK;
}

int v2x_dqn_targets(const float* q, const float* q_next, const int32_t* action, const double* reward, double gamma,
                    int32_t n_graphs, int32_t n_nodes, int32_t n_channels, float* y_out, void* stream) {
  v2x_model* nullm = nullptr;
  if (!q || !q_next || !action || !reward || !y_out || n_graphs <= 0 || n_nodes <= 0 || n_channels <= 0)
    FAIL(nullm, V2X_EINVAL, "dqn_targets: bad argument");
  const int n_rows = n_graphs * n_nodes;
  hipLaunchKernelGGL(k_dqn_targets, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, q_next, action,
                     reward, gamma, n_rows, n_nodes, n_channels, y_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) FAIL(nullm, V2X_EHIP, "dqn_targets launch failed: %s", hipGetErrorString(e));
  return V2X_OK;
}

// ---------------------------------------------------------------------------- dict payload -> packed host batch
int v2x_pack_feed(const v2x_feed* f, int check_kron, float* xe_out, int32_t* row_ptr_out, int32_t* col_idx_out,
                  float* nbr_out, int32_t* info_out) {
  v2x_model* nullm = nullptr;
  if (!f || !xe_out || !row_ptr_out || !col_idx_out || !info_out || !f->node || !f->edge || !f->is_f64 || !f->adjacency)
    FAIL(nullm, V2X_EINVAL, "pack_feed: null argument");
  if (f->n_graphs <= 0 || f->n_nodes <= 0 || f->feat_dim <= 0 || f->node_in <= 0 || f->edge_in < 0 ||
      f->node_in + f->edge_in > XE)
    FAIL(nullm, V2X_EINVAL, "pack_feed: bad sizes (node_in + edge_in must fit the %d-wide packed row)", XE);
  if ((int64_t)f->n_graphs * f->n_nodes * f->n_nodes > INT32_MAX) FAIL(nullm, V2X_EINVAL, "pack_feed: batch too large");
  for (int k = 0; k < f->n_nodes; ++k)
    if (!f->node[k] || !f->edge[k] || (f->nbr && !f->nbr[k])) FAIL(nullm, V2X_EINVAL, "pack_feed: null input array");
  v2x_host::FeedView v{f->n_graphs, f->n_nodes, f->feat_dim, f->node_in, f->edge_in, f->node, f->edge, f->nbr, f->is_f64, f->adjacency};
  int bad = -1;
  const int rc = v2x_host::pack_feed(v, check_kron != 0, xe_out, row_ptr_out, col_idx_out, nbr_out, info_out, &bad);
  if (rc == v2x_host::PACK_NOT_01)
    FAIL(nullm, V2X_EINVAL, "adjacency entries must be 0 or 1 (the engine aggregates unweighted edges); sample %d", bad);
  if (rc == v2x_host::PACK_NOT_KRON) FAIL(nullm, V2X_EINVAL, "Adjacency_Matrix is not kron(Adj, I_F) (sample %d)", bad);
  return V2X_OK;
}

// ---------------------------------------------------------------------------- contract checks
int v2x_validate_batch(v2x_model* m, const v2x_batch* b, int32_t n_nodes, void* stream) {
  if (!b || b->n_graphs <= 0 || b->n_rows <= 0 || !b->row_ptr || (b->n_edges > 0 && !b->col_idx) || b->max_nodes <= 0 ||
      b->max_edges < 0)
    FAIL(m, V2X_EINVAL, "validate_batch: null pointer or non-positive size");
  if (m) n_nodes = m->N;
  if (!b->graph_off && (n_nodes <= 0 || (int64_t)b->n_graphs * n_nodes != b->n_rows))
    FAIL(m, V2X_EINVAL, "validate_batch: n_rows (%d) != n_graphs*n_nodes (%d*%d)", b->n_rows, b->n_graphs, n_nodes);
  if (!b->on_device) return validate_host_batch(m, b, n_nodes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_validate_batch, dim3(std::min(b->n_graphs, 4096)), dim3(256), 0, st, b->graph_off, b->row_ptr, b->col_idx,
                     b->n_graphs, b->n_rows, b->n_edges, n_nodes, b->max_nodes, b->max_edges, flag_dev_of(m));
  HIPCHK(m, hipGetLastError());
  HIPCHK(m, hipStreamSynchronize(st));
  return check_flag(m);
}

int v2x_reset_exchange(v2x_model* m) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  HIPCHK(m, hipSetDevice(m->cfg.device));
  return reset_exchange(m);
}

void* v2x_debug_exchange_counters(v2x_model* m) { return m ? (void*)m->small_sync : nullptr; }
void* v2x_debug_split_counters(v2x_model* m, int32_t* n_tiles) {
  if (n_tiles) *n_tiles = m ? m->xchg_cap_tiles : 0;
  return m ? (void*)m->xchg_sync : nullptr;
}

int v2x_check_errors(v2x_model* m, void* stream) {
  HIPCHK(m, hipStreamSynchronize((hipStream_t)stream));
  return check_flag(m);
}

// ---------------------------------------------------------------------------- measurement
int v2x_path_info(v2x_model* m, const v2x_batch* b, char* out, int cap) {
  if (!m || !b || !out || cap < 16) FAIL(m, V2X_EINVAL, "path_info: bad argument");
  DevBatch d;
  memset(&d, 0, sizeof(d));
  d.B = b->n_graphs; d.R = b->n_rows; d.E = b->n_edges; d.max_nodes = b->max_nodes; d.max_edges = b->max_edges;
  d.goff = b->graph_off; d.nbr = b->nbr_init;
  const bool fused = fused_path(m, d);
  const int split = fused_split(m, d);
  // "edge-bitset-walk": what the fused kernels do with the edge index (N <= 32) -- every CSR row becomes a 32-bit set, a wave
  // walks the graph's rows in LDS once and adds each to the slots whose set holds it (sparse lanes walk their set bits only);
  // "edge-gather": the per-edge CSR gather / segment sum of k_agg (layer-wise path, any N)
  const char* agg = fused ? (split <= 1 && fused_compl(m, d) ? "complement" : "edge-bitset-walk")
                          : (use_dense_agg(d, m->F) ? "dense(complement-or-mfma-per-graph)" : "edge-gather");
  char gl[48];
  if (fused && split > 1) snprintf(gl, sizeof(gl), "fused(split%d)", split);
  else snprintf(gl, sizeof(gl), "%s", fused ? "fused" : (ragged_fused_path(m, d) ? "fused(ragged)" : "layerwise"));
  // dense0_dw: which launch takes Dense-0's weight gradient in a fit step -- the MLP launch itself, or (small batches) a role of
  // the graph layers' weight-gradient launch (dense0_rides)
  const bool d0 = mlp_wg_path(m) && !m->cfg.variable_graphs && dense0_rides(m, d, idx_map(m, d, Range{0, d.B}));
  snprintf(out, cap, "graph_layers=%s aggregation=%s mlp=%s handoff=%s dense0_dw=%s", gl, agg,
           (d0 && mlp_stream_rides(m, d, idx_map(m, d, Range{0, d.B}))) ? "stream" : (mlp_wg_path(m) ? "train_wg" : (mlp_fused_training(m) ? "train" : "fwd+bwd")),
           frag_layout(m, d, Range{0, d.B}) ? "fragment-major" : "row-major",
           d0 ? "k_wgrad" : (mlp_wg_path(m) ? "k_mlp_train_wg" : (is_wide(m) ? "k_wide_wgrad" : "k_wgrad")));
  return V2X_OK;
}

int v2x_debug_phase_stamps(v2x_model* m, int64_t* out, int n) {
  if (!m || !out) FAIL(m, V2X_EINVAL, "null argument");
  if (!m->ts_buf) FAIL(m, V2X_ESTATE, "phase stamps need V2X_FUSED_TS=1 when the model is created");
  HIPCHK(m, hipDeviceSynchronize());
  HIPCHK(m, hipMemcpy(out, m->ts_buf, (size_t)std::min(n, 4 * 8 * 64) * 8, hipMemcpyDeviceToHost));
  return V2X_OK;
}

int v2x_debug_ragged_plan(v2x_model* m, int32_t* out, int n) {
  if (!m || !out || n < 0) FAIL(m, V2X_EINVAL, "null argument");
  if (!m->plan_buf.p || m->plan_len <= 0) FAIL(m, V2X_ESTATE, "no ragged fused forward has run on this model");
  HIPCHK(m, hipDeviceSynchronize());
  HIPCHK(m, hipMemcpy(out, m->plan_buf.p, (size_t)std::min(n, m->plan_len) * 4, hipMemcpyDeviceToHost));
  return m->plan_len;
}

int v2x_profile_enable(v2x_model* m, int enable) {
  if (!m) FAIL(m, V2X_EINVAL, "null model");
  m->prof = enable != 0;
  if (!enable) {
    for (auto& r : m->prof_recs) { hipEventDestroy(r.ev0); hipEventDestroy(r.ev1); }
    m->prof_recs.clear();
  }
  return V2X_OK;
}

int v2x_profile_read(v2x_model* m, char* names_out, int names_cap, double* ms_out, int64_t* calls_out, int max_entries) {
  if (!m || !names_out || !ms_out || !calls_out) FAIL(m, V2X_EINVAL, "profile_read: null argument");
  HIPCHK(m, hipDeviceSynchronize());
  std::vector<double> ms(m->prof_names.size(), 0.0);
  std::vector<int64_t> calls(m->prof_names.size(), 0);
  for (auto& r : m->prof_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.ev0, r.ev1) == hipSuccess) { ms[r.id] += t; calls[r.id] += 1; }
    hipEventDestroy(r.ev0); hipEventDestroy(r.ev1);
  }
  m->prof_recs.clear();
  std::string names;
  int n = 0;
  for (size_t i = 0; i < m->prof_names.size() && n < max_entries; ++i) {
    if (!calls[i]) continue;
    if ((int)(names.size() + m->prof_names[i].size() + 2) > names_cap) break;
    names += m->prof_names[i]; names += '\n';
    ms_out[n] = ms[i]; calls_out[n] = calls[i];
    ++n;
  }
  snprintf(names_out, names_cap, "%s", names.c_str());
  return n;
}

}  // extern "C"
