/*
 * v2xgnn.h -- C ABI of the MI355X (gfx950) GNN message-passing engine that replaces the
 * Keras/TF1 Q-network of the reference (`/root/reference/BS_brain.py`).
 *
 * This is the drop-in boundary.  The reference has no FFI of its own
 * (it is pure Python on Keras); the entry points below are what a `ctypes` binding of the
 * reference's `BS` class needs, one per reference call site:
 *
 *   v2x_create / v2x_destroy      <- BS._create_model            BS_brain.py:108-216
 *   v2x_forward                   <- Model.predict               BS_brain.py:225-235
 *   v2x_train_step                <- Model.fit (1 step)          BS_brain.py:218-223
 *   v2x_copy_weights              <- BS.update_target_model      BS_brain.py:237-239
 *   v2x_get_weights/set_weights   <- get_weights / set_weights / save_weights / load_weights
 *                                                                BS_brain.py:239,863,869,1254
 *   v2x_gather_rows / v2x_dqn_targets / v2x_dqn_step
 *                                 <- Agent.replay batching, target rule, whole step  BS_brain.py:555-748
 *   v2x_agg_* / v2x_node_update_* / v2x_mlp_* / v2x_adam_step
 *                                 <- the implicit TF op set of GNNLayer.call (:44-51),
 *                                    AggLayer.call (:69-76), Dense (:176-179), huber (:86-87),
 *                                    Adam (:212); exported so each kernel is parity-testable
 *                                    on its own.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns 0 on success or a negative V2X_E* code;
 *     `v2x_last_error(model)` (or `v2x_last_error(NULL)` for create failures) gives text.
 *   - all arithmetic is fp32 (Keras floatx), edge indices int32.
 *   - pointers marked [dev] are device (HBM) pointers, [host] host pointers, [any] either,
 *     selected by the `on_device` flag next to them.  The caller owns every buffer it
 *     passes; the engine never retains input pointers beyond the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream except where host buffers are read or written.
 *   - one model handle per (network, device); a handle is not thread-safe; distinct
 *     handles are independent (online and target networks are two handles).
 *
 * Data layout (DESIGN.md "Data layout in HBM")
 *   A batch of B graphs is R node rows, graph-major; node order inside a graph is the
 *   reference's D1..DN order.  Features are packed per row as
 *       xe[R][16] = [ node features (Dn = 2C+1) | edge features (De = C) | zero pad ]
 *   (the reference's `D{k}_Node_Input` / `D{k}_Edge_Input`, BS_brain.py:461-467).
 *   Adjacency is CSR by DESTINATION row: sources of global row q are the graph-local node
 *   ids col_idx[row_ptr[q] .. row_ptr[q+1]) , ascending, no duplicates; this is
 *   Adj[p,q] == 1  (BS_brain.py:441-445) and  agg_q = sum_p Adj[p,q] h_p  (:72-76).
 *   graph_off[B+1] gives the first row of every graph (NULL => fixed n_nodes per graph).
 *
 * Flat parameter layout (v2x_get_weights / v2x_set_weights / gradient buffer), fp32:
 *   stage-major, slot-minor.  S = n_nodes weight sets (reference: one per node) or 1 when
 *   share_weights.  For every layer, for every slot:  W[K][N_out] row-major, then bias[N_out].
 *     GNN stage 0      K rows = [ x(Dn) | e(De) | neighbor(F) ]            (W1 ; W2 ; W3 of :121)
 *     GNN stage s>=1   K rows = [ h(F) | x(Dn) | e(De) | agg(F) ]         (W1 ; W2 ; W3 of :154/:161)
 *     Dense 0          K rows = [ h(F) | x(Dn) | agg(F) ]   N_out = 80     (:176, rows permuted
 *                                                                          from Keras' [x|h|agg])
 *     Dense 1..3       [80][40], [40][20], [20][C]                         (:177-179)
 */
#ifndef V2XGNN_H
#define V2XGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2X_OK            0
#define V2X_EINVAL       -1   /* bad argument / unsupported configuration */
#define V2X_EHIP         -2   /* a HIP runtime call failed                */
#define V2X_ENOMEM       -3
#define V2X_ESTATE       -4   /* call order violated (e.g. backward before forward) */

#define V2X_XE_WIDTH     16   /* packed [x|e|pad] row width */

typedef struct v2x_model v2x_model;

typedef struct v2x_config {
  int32_t n_nodes;        /* N: nodes per graph (num_D2D, BS_brain.py:95); >=1             */
  int32_t n_channels;     /* C: num_CH (:97); this build supports C == 4                     */
  int32_t feat_dim;       /* F: num_Feedback (:98); 16, 32, 64 (register-chained path), 128 or
                             256 (LDS-tiled wide path) in this build                         */
  int32_t n_mp_layers;    /* L: message-passing stages after the embed (reference: 2)        */
  int32_t share_weights;  /* 0 = one weight set per node slot (reference), 1 = shared        */
  int32_t variable_graphs;/* 1 = graphs of different sizes (needs share_weights)             */
  int32_t device;         /* HIP device ordinal                                              */
  int32_t use_graph;      /* 1 = capture/replay the step as a hipGraph when shapes repeat    */
  float   lr, beta1, beta2, eps;   /* Keras Adam (:212): 1e-3, 0.5, 0.999, 1e-7             */
} v2x_config;

typedef struct v2x_batch {
  int32_t n_graphs;        /* B */
  int32_t n_rows;          /* R = sum of nodes                                              */
  int32_t n_edges;         /* E = row_ptr[R]                                                */
  int32_t max_nodes;       /* max nodes of any graph in the batch                           */
  int32_t max_edges;       /* max edges of any graph in the batch                           */
  int32_t on_device;       /* 1: all pointers below are [dev]; 0: [host] (copied per call)  */
  const float*   xe;        /* [R][16]                                                      */
  const float*   nbr_init;  /* [R][F] or NULL (reference always feeds zeros, :478-490)      */
  const int32_t* graph_off; /* [B+1] or NULL when every graph has n_nodes rows              */
  const int32_t* row_ptr;   /* [R+1] global edge offsets                                    */
  const int32_t* col_idx;   /* [E] graph-local source node                                  */
} v2x_batch;

/* ---- model lifetime -------------------------------------------------------------------- */
int  v2x_create(const v2x_config* cfg, v2x_model** out);
void v2x_destroy(v2x_model* m);
const char* v2x_last_error(const v2x_model* m);
const char* v2x_version(void);

/* ---- parameters ------------------------------------------------------------------------ */
int64_t v2x_param_count(const v2x_model* m);
int  v2x_get_weights(v2x_model* m, float* host_out, void* stream);        /* [host] P floats */
int  v2x_set_weights(v2x_model* m, const float* host_in, void* stream);   /* [host] P floats */
int  v2x_copy_weights(v2x_model* dst, const v2x_model* src, void* stream);/* device-to-device */
int  v2x_get_optimizer_state(v2x_model* m, float* host_m, float* host_v, int64_t* iterations, void* stream);
int  v2x_set_optimizer_state(v2x_model* m, const float* host_m, const float* host_v, int64_t iterations, void* stream);
/* device pointers of the flat fp32 buffers (length v2x_param_count): for RCCL all-reduce of
 * the gradient by the host framework, and for zero-copy inspection.                        */
float* v2x_param_ptr(v2x_model* m);
float* v2x_grad_ptr(v2x_model* m);

/* ---- the hot path ---------------------------------------------------------------------- */
/* forward only: q_out[R][C]  (Model.predict, BS_brain.py:225-231).  Batches of at most 256 node
 * rows of fixed-size graphs (the rollout predict, BS_brain.py:336,1108,1394: one graph per call) run as
 * ONE launch (csrc/kernels_small.hpp); nothing is saved for a backward pass, which is only ever
 * driven by v2x_forward_backward / v2x_train_step / v2x_dqn_step (they run their own forward).      */
int  v2x_forward(v2x_model* m, const v2x_batch* b, float* q_out, int q_on_device, void* stream);

/* v2x_forward as a plain callback `int (*)(void*)` for host code that takes one: the closure holds the arguments (the batch
 * descriptor may point into page-locked host buffers the caller refills between calls, on_device = 1, see
 * v2x_device_addressable; q_on_device = 0 copies Q back and synchronises).  The rollout of ONE simulator as a single
 * library call (include/v2xsim.h, v2xsim_rollout: BS_brain.py:308-352 inside :409-553) scores its observations through it.   */
typedef struct v2x_forward_closure {
  v2x_model* m; v2x_batch b; float* q_out; int32_t q_on_device; int32_t pad_; void* stream;
} v2x_forward_closure;
int  v2x_forward_call(void* closure);

/* one fit step = forward + Huber + backward + Keras-Adam (Model.fit, BS_brain.py:218-223).
 *   y[R][C] targets;  loss_out[n_nodes] per-output Huber means (History 'D{k}_Decide_Output_loss'),
 *   may be NULL.  n_graphs_global: B of the GLOBAL batch (== b->n_graphs on one GPU): the
 *   Huber mean is taken over it, so per-rank gradients SUM to the global-batch gradient.   */
int  v2x_train_step(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device,
                    int32_t n_graphs_global, float* loss_out, int loss_on_device, void* stream);

/* the same step split for data parallelism: (1) forward+backward leaves the local gradient
 * in v2x_grad_ptr(); the host all-reduces (sum) it over ranks; (2) apply the Adam update.  */
int  v2x_forward_backward(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device,
                          int32_t n_graphs_global, float* loss_out, int loss_on_device, void* stream);
int  v2x_apply_gradients(v2x_model* m, void* stream);
/* the same forward+backward in v2x_grad_bucket_count(m) calls ("phases"), for overlapping the gradient all-reduce with the
 * backward pass: after phase k, bucket k of the flat gradient is final and the host may start its collective while the
 * later phases compute.  v2x_grad_bucket gives a bucket's length and its offset (floats) inside v2x_grad_ptr().
 *   feat_dim <= 64 (graph-major fused kernels): 2 phases -- [the Dense layers] (tail of the flat layout), [the graph layers]
 *   feat_dim >= 128 (one weight-gradient launch per layer): L + 2 phases -- [the Dense layers], [stage L], ..., [stage 1],
 *     [embed]: 4.6 + 3 x 13.5 + 6.9 M floats at configs[3] ("bandwidth-bound, overlappable with the tail
 *     of backward")
 * loss_out is written by the last phase.  Phases must be called in order, 0 first.                                        */
int  v2x_forward_backward_phase(v2x_model* m, const v2x_batch* b, const float* y, int y_on_device,
                                int32_t n_graphs_global, int phase, float* loss_out, int loss_on_device, void* stream);
int  v2x_grad_bucket_count(const v2x_model* m);
int64_t v2x_grad_bucket(const v2x_model* m, int bucket, int64_t* offset);
/* Keras Adam on parameters [offset, offset + count) only (float4-aligned), from the gradient buffer: the optimizer step of
 * a rank that owns a 1 / G slice of every bucket after a reduce-scatter (the host all-gathers the parameters afterwards).
 * advance_iteration != 0 on the first call of a step (Adam's t), 0 on the others.                                       */
int  v2x_apply_gradients_range(v2x_model* m, int64_t offset, int64_t count, int advance_iteration, void* stream);

/* ---- per-kernel entry points (parity tests; all pointers [dev]) ------------------------ */
/* AggLayer.call forward: out[q] = sum_{p in N(q)} h[p]            (BS_brain.py:69-76)      */
int  v2x_agg_fwd(const v2x_batch* b, int32_t n_nodes, int32_t feat_dim,
                 const float* h, float* out, void* stream);
/* its transpose (backward):   out[p] = sum_{q : p in N(q)} g[q]                            */
int  v2x_agg_bwd(const v2x_batch* b, int32_t n_nodes, int32_t feat_dim,
                 const float* g, float* out, void* stream);
/* GNNLayer.call of stage `stage` with the model's weights (BS_brain.py:44-51):
 *   out = act( [h_prev|x]W1 + e W2 + agg_prev W3 + b ), act = relu for stage < L.          */
int  v2x_node_update_fwd(v2x_model* m, int32_t stage, int32_t n_rows, const float* xe,
                         const float* h_prev, const float* agg_prev, float* out, void* stream);
/* backward of the same: given dpre[R][F] (gradient at the pre-activation) writes
 * dh_prev[R][F], dagg_prev[R][F] (either may be NULL for stage 0) and ACCUMULATES nothing:
 * the layer's weight gradient is written to grad_out (flat layout, only this layer's range). */
int  v2x_node_update_bwd(v2x_model* m, int32_t stage, int32_t n_rows, const float* xe,
                         const float* h_prev, const float* agg_prev, const float* dpre,
                         float* dh_prev, float* dagg_prev, float* grad_out, void* stream);
/* decision MLP (Dense 80-40-20-C, :176-179) forward and Huber+backward.                    */
int  v2x_mlp_fwd(v2x_model* m, int32_t n_rows, const float* xe, const float* h, const float* agg,
                 float* q_out, void* stream);
int  v2x_mlp_huber_bwd(v2x_model* m, int32_t n_rows, int32_t n_graphs_global, const float* xe,
                       const float* h, const float* agg, const float* y,
                       float* dh, float* dagg, float* grad_out, float* loss_out, void* stream);
/* Keras Adam on arbitrary flat device buffers (BS_brain.py:212)     */
int  v2x_adam_step(float* param, const float* grad, float* mom, float* vel, int64_t n,
                   int64_t iteration /* 1-based t */, float lr, float beta1, float beta2, float eps,
                   void* stream);

/* ---- DQN replay glue on device ------------------------------------------------------------
 * Counterparts of the minibatch assembly (BS_brain.py:573-640: per-sample Python loops filling the
 * 13 input arrays) and of the target rule (BS_brain.py:670-692) of Agent.replay, for transitions
 * that stay resident in HBM.  All pointers [dev].                                              */
/* 1 when the device can dereference p as it stands: device or managed memory, or page-locked host memory that is mapped
 * at the SAME address (hipHostMalloc defaults under unified addressing: what the zero-copy predict and the replay's index
 * ring hand to kernels, rl/agent.py, rl/replay.py); 0 for pageable host memory or a mapping at another address; a negative
 * V2X_E* code when the runtime cannot say.  Callers check once per buffer, before they set on_device = 1 on host arrays.    */
int  v2x_device_addressable(const void* p);
/* dst[i][0..row_bytes) = src[idx[i]][0..row_bytes)   (row_bytes a multiple of 4)               */
int  v2x_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t n_idx, int64_t row_bytes,
                     void* stream);
/* up to 8 such gathers by the SAME index list as one launch (a replay minibatch: s, s', actions, rewards, CSR sources):
 * dst[j][i][0..row_bytes[j]) = src[j][idx[i]][0..row_bytes[j]).  src / dst / row_bytes: [host] arrays of n_jobs entries.       */
int  v2x_gather_rows_multi(int32_t n_jobs, const void* const* src, void* const* dst, const int64_t* row_bytes,
                           const int32_t* idx, int64_t n_idx, void* stream);
/* Q statistics of the fitted targets y[n_graphs][n_nodes][n_channels] (BS_brain.py:730-746) as float64 sums per link:
 * out[0][k] = sum over samples and channels, out[1][k] = sum over samples of the per-sample maximum; the caller divides
 * (by n_graphs * n_channels, by n_graphs).  out: [dev] [2][n_nodes] doubles.  Deterministic (fixed summation order).       */
int  v2x_q_stats(const float* y, int32_t n_graphs, int32_t n_nodes, int32_t n_channels, double* out, void* stream);
/* y = q (online net on s), except y[b][k][action[b][k]] = reward[b] + gamma * max_c q_next[b][k][c]
 * (q_next: target net on s'); evaluated like the reference's numpy-1.x scalar expression (float64
 * product and sum, rounded to fp32 once).  q, q_next, y: [n_graphs*n_nodes][n_channels]; action: [n_graphs][n_nodes];
 * reward: [n_graphs] double.                                                                    */
int  v2x_dqn_targets(const float* q, const float* q_next, const int32_t* action, const double* reward,
                     double gamma, int32_t n_graphs, int32_t n_nodes, int32_t n_channels, float* y_out,
                     void* stream);

/* One whole replay step (BS_brain.py:664-728: predict on s, predict(target) on s', target rule, train_dnn) as a
 * single call on device-resident batches: the graph layers of the online network run ONCE (their activations feed
 * the backward pass), where predict + fit would run them twice.  y_out: optional [dev] [n_rows][C] buffer that
 * receives the training targets (for the caller's Q statistics, BS_brain.py:730-746); loss_out: per-output Huber
 * means like v2x_train_step.                                                                                   */
int  v2x_dqn_step(v2x_model* online, v2x_model* target, const v2x_batch* s, const v2x_batch* s_next,
                  const int32_t* action, const double* reward, double gamma, int32_t n_graphs_global,
                  float* y_out, float* loss_out, int loss_on_device, void* stream);

/* ---- the dict payload of the reference -> a packed host batch ---------------------------------
 * Model.predict / Model.fit of the reference take, per call, 3 N arrays 'D{k}_Node_Input' [B][Dn], 'D{k}_Edge_Input'
 * [B][De], 'D{k}_Neighbor_Input' [B][F] and the dense 'Adjacency_Matrix' [B][N F][N F] = kron(Adj, I_F)
 * (BS_brain.py:492-504, :603, :642-651, :704-716).  v2x_pack_feed turns that payload into the arrays of a host v2x_batch
 * in one pass of compiled host code (csrc/host_pack.hpp; no GPU involved): xe_out [B N][16], row_ptr_out [B N + 1],
 * col_idx_out [capacity B N N], nbr_out [B N][F] (written only when some Neighbor_Input entry is non-zero -- the
 * reference always feeds zeros, :478-490), info_out = {n_edges, max_edges, neighbour input non-zero}.
 * Every array is C-contiguous float32 or float64 (is_f64[3 N + 1]: node[0..N), edge[0..N), nbr[0..N), adjacency).
 * check_kron != 0: V2X_EINVAL unless the adjacency is exactly kron(Adj, I_F); entries other than 0 / 1 are always
 * V2X_EINVAL (the engine aggregates unweighted edges).  Error text: v2x_last_error(NULL).                              */
typedef struct v2x_feed {
  int32_t n_graphs, n_nodes, feat_dim, node_in, edge_in;
  const void* const* node;        /* [n_nodes] pointers */
  const void* const* edge;        /* [n_nodes] pointers */
  const void* const* nbr;         /* [n_nodes] pointers, or NULL */
  const uint8_t* is_f64;          /* [3 * n_nodes + 1] */
  const void* adjacency;
} v2x_feed;
int  v2x_pack_feed(const v2x_feed* feed, int check_kron, float* xe_out, int32_t* row_ptr_out, int32_t* col_idx_out,
                   float* nbr_out, int32_t* info_out);

/* ---- contract checks -------------------------------------------------------------------
 * The kernels size their LDS tiles from max_nodes / max_edges and rely on the CSR contract above (sources inside
 * their graph, strictly ascending => no duplicates).  HOST batches are checked on every call before anything is copied
 * (V2X_EINVAL).  DEVICE batches are the caller's responsibility: v2x_validate_batch checks one (synchronises `stream`;
 * `m` may be NULL, then n_nodes is used for fixed-size graphs).  Independently, a workgroup that finds a graph larger than
 * its LDS tile stages nothing and raises a flag; the flag is reported (V2X_EINVAL, results invalid) by the next call
 * that synchronises with the host (host-side q / loss outputs) or by v2x_check_errors.                              */
int  v2x_validate_batch(v2x_model* m, const v2x_batch* b, int32_t n_nodes, void* stream);
int  v2x_check_errors(v2x_model* m, void* stream);
/* The one-launch rollout predict (csrc/kernels_small.hpp) hands node rows between its workgroups through an exchange buffer
 * whose only state across launches is a per-graph departure count.  A launch that was aborted half-way leaves that state out
 * of step; the next predict then does not hang: its polls are bounded, it raises a flag and the synchronising call returns
 * V2X_ESTATE after re-arming the exchange by itself.  v2x_reset_exchange does the same re-arming on request (synchronises
 * the device).  v2x_debug_exchange_counters: [dev] pointer to the 256 64-bit departure counters (tests corrupt one on purpose).
 * The split-tile fused graph layers (csrc/kernels_fused_split.hpp: K workgroups per 16-graph tile for batches that leave most
 * of the chip idle) hand stage rows between a tile's workgroups the same way, with one 64-bit launch counter per tile; the
 * same bounded polls, the same V2X_ESTATE + self re-arming, and v2x_reset_exchange re-arms that exchange as well.           */
int  v2x_reset_exchange(v2x_model* m);
void* v2x_debug_exchange_counters(v2x_model* m);
/* [dev] pointer to the split-tile exchange's per-tile 64-bit launch counters (*n_tiles of them; NULL before the first split-tile
 * launch of the model): tests push one out of step on purpose.                                                                    */
void* v2x_debug_split_counters(v2x_model* m, int32_t* n_tiles);

/* ---- measurement ------------------------------------------------------------------------ */
/* When enabled, every kernel launch of this model is bracketed by HIP events on its stream
 * (eager, no graph); v2x_profile_read returns per-kernel-name call counts and total ms.    */
int  v2x_profile_enable(v2x_model* m, int enable);
/* Which kernels a fit step of `b` runs on this model, as text ("graph_layers=fused aggregation=complement mlp=train_wg
 * handoff=fragment-major"): the aggregation is the general edge-index gather / segment sum of AggLayer.call
 * (BS_brain.py:69-76) unless the batch is dense enough for one of its rewritings -- through the complement in the fused
 * graph-layer kernels ("complement"), per graph through the complement or as an MFMA product with adjacency bit masks for large dense graphs
 * ("dense(complement-or-mfma-per-graph)").
 * Only sizes and null-ness of the batch pointers are read.  bench.py prints it in `config.aggregation`.              */
int  v2x_path_info(v2x_model* m, const v2x_batch* b, char* out, int cap);
/* Models created with V2X_FUSED_TS=1 in the environment run a measurement build of the fused forward kernel in which
 * workgroup 7 writes 100 MHz time stamps at its phase boundaries: out[wave * 64 + mark], n <= 512 entries.            */
int  v2x_debug_phase_stamps(v2x_model* m, int64_t* out, int n);
/* Tests: the work plan of the last ragged fused forward (csrc/kernels_ragged.hpp) -- out[w] = first graph of workgroup w for
 * w = 0 .. n - 1 (entries past the plan's length are left alone); returns the plan's length (launched workgroups + 1) or a
 * negative error code.  [host] out. */
int  v2x_debug_ragged_plan(v2x_model* m, int32_t* out, int n);
int  v2x_profile_read(v2x_model* m, char* names_out, int names_cap, double* ms_out, int64_t* calls_out,
                      int max_entries);   /* returns number of entries, names '\n'-separated */

#ifdef __cplusplus
}
#endif
#endif /* V2XGNN_H */
