/* libv2xsim.so -- the array arithmetic of one batched simulator step, the environments spread over a pool of threads.
 *
 * Counterpart of rl/batched_env.py's numpy expressions (which restate /root/reference/Environment.py:94-146 path loss,
 * :378-393 shadowing, :395-406 Rayleigh fast fading, :408-493 rates / interference and BS_brain.py:389-467 observation)
 * for E independent environments of n vehicles / links.  The random streams stay in Python (MT19937 per environment with
 * the stdlib's draw order, rl/mtstream.py): this library receives the UNIFORMS of a step and turns them into Gaussians
 * in random.gauss order (cos value, then the sin value of the same pair).  Same formulas, same operation order as the
 * numpy code; results agree to the last bits of libm (the tests compare at 1e-12).
 * Plain C, no Python API: bound with ctypes (rl/native_sim.py).  gcc -O2 -pthread -shared -fPIC v2xsim.c -lm         */
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/v2xsim.h"

#define TWOPI 6.283185307179586476925286766559

static const double V2V_H = 1.5, FC = 2.0, V2V_DECORR = 10.0, V2V_SHADOW_STD = 3.0;
static const double V2I_H_BS = 25.0, V2I_H_MS = 1.5, V2I_DECORR = 50.0, V2I_SHADOW_STD = 8.0;
static const double BS_X = 750.0 / 2, BS_Y = 1299.0 / 2;

int v2xsim_abi(void) { return 3; }
/* ---- the thread pool --------------------------------------------------------------------------------------------------
 * Every entry point is "for each environment e: f(e)".  The loops run on a pool of threads that SLEEP between jobs (condition
 * variable): an OpenMP team spins for a while after each parallel region, and on the MI355X boxes the process may use 16 CPUs'
 * worth of time per 100 ms (cgroup cpu.max) -- two spinning teams next to the GPU runtime's threads exhausted that and the
 * kernel parked the whole process for 25-50 ms, once or twice per hundred train steps (tools/prof_rl_sections.py, rounds 4-5).
 * par_for: the caller takes part and returns when all environments are done.  par_start / par_wait: the pool alone works
 * on the job (the look-ahead step below); while such a job is in flight par_for runs on the caller alone.
 * Tickets carry the job's generation, so a worker that wakes up late can never take an index of a later job with an
 * earlier job's function.                                                                                               */
typedef void (*env_fn)(int e, void* ctx);
#define POOL_MAX 64
static struct {
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  pthread_t th[POOL_MAX];
  int n_started;
  env_fn fn; void* ctx; int n; int n_workers;       /* the current job (read under mu) */
  uint32_t gen;
  _Atomic uint64_t ticket;                          /* (gen << 32) | next index */
  _Atomic int remaining;
  int async_busy;                                   /* a par_start job is in flight (or done and not yet waited for) */
  int async_id;                                     /* its ticket */
  int sync_busy;                                    /* a par_for job of ANOTHER caller is in flight (ctypes releases the GIL: two
                                                       host threads may step two simulators at once): later callers run alone */
} P = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER };
static int g_threads = 1;            /* threads a loop may use, the caller included (v2xsim_set_threads) */

static int take(uint32_t gen, int n) {              /* next index of job `gen`, or -1 */
  uint64_t t = atomic_load(&P.ticket);
  for (;;) {
    if ((uint32_t)(t >> 32) != gen || (int)(uint32_t)t >= n) return -1;
    if (atomic_compare_exchange_weak(&P.ticket, &t, t + 1)) return (int)(uint32_t)t;
  }
}
static void finish_one(void) {
  if (atomic_fetch_sub(&P.remaining, 1) == 1) {
    pthread_mutex_lock(&P.mu);
    pthread_cond_broadcast(&P.cv_done);
    pthread_mutex_unlock(&P.mu);
  }
}
static void* pool_main(void* arg) {
  const int id = (int)(intptr_t)arg;
  uint32_t seen = 0;
  pthread_mutex_lock(&P.mu);
  for (;;) {
    while (P.gen == seen) pthread_cond_wait(&P.cv_work, &P.mu);
    seen = P.gen;
    const env_fn fn = P.fn; void* ctx = P.ctx; const int n = P.n, allowed = P.n_workers;
    pthread_mutex_unlock(&P.mu);
    if (id < allowed)
      for (int e; (e = take(seen, n)) >= 0;) { fn(e, ctx); finish_one(); }
    pthread_mutex_lock(&P.mu);
  }
  return 0;
}
static void pool_after_fork(void) {                 /* the threads do not exist in a forked child: start over */
  pthread_mutex_init(&P.mu, 0);
  pthread_cond_init(&P.cv_work, 0);
  pthread_cond_init(&P.cv_done, 0);
  P.n_started = 0;
  P.async_busy = 0;
  P.sync_busy = 0;
  atomic_store(&P.remaining, 0);
}
/* post a job for `workers` pool threads (mu held) */
static void post(env_fn fn, void* ctx, int n, int workers) {
  static int atfork_set = 0;
  if (!atfork_set) { pthread_atfork(0, 0, pool_after_fork); atfork_set = 1; }
  if (workers > POOL_MAX) workers = POOL_MAX;
  while (P.n_started < workers) {
    if (pthread_create(&P.th[P.n_started], 0, pool_main, (void*)(intptr_t)P.n_started) != 0) break;
    pthread_detach(P.th[P.n_started]);
    ++P.n_started;
  }
  P.fn = fn; P.ctx = ctx; P.n = n; P.n_workers = workers < P.n_started ? workers : P.n_started;
  ++P.gen;
  atomic_store(&P.remaining, n);
  atomic_store(&P.ticket, (uint64_t)P.gen << 32);
  if (P.n_workers > 0) pthread_cond_broadcast(&P.cv_work);
}
static void par_for(int n, env_fn fn, void* ctx) {
  if (n <= 0) return;
  pthread_mutex_lock(&P.mu);
  if ((P.async_busy && atomic_load(&P.remaining) > 0) || P.sync_busy || g_threads <= 1 || n == 1) {
    /* the pool is working ahead, or on another caller's loop (the job slot is ONE: posting over it would strand that caller's
       environments), or is not wanted: the caller alone */
    pthread_mutex_unlock(&P.mu);
    for (int e = 0; e < n; ++e) fn(e, ctx);
    return;
  }
  P.sync_busy = 1;
  post(fn, ctx, n, (g_threads < n ? g_threads : n) - 1);
  const uint32_t gen = P.gen;
  pthread_mutex_unlock(&P.mu);
  for (int e; (e = take(gen, n)) >= 0;) { fn(e, ctx); atomic_fetch_sub(&P.remaining, 1); }
  for (int spin = 0; spin < 4000 && atomic_load(&P.remaining) > 0; ++spin) __builtin_ia32_pause();
  pthread_mutex_lock(&P.mu);
  while (atomic_load(&P.remaining) > 0) pthread_cond_wait(&P.cv_done, &P.mu);
  P.sync_busy = 0;
  pthread_mutex_unlock(&P.mu);
}
/* > 0: started, the job's ticket; -1: another job is still RUNNING (one that is done but was never waited for -- its owner
 * forgot it or died -- is simply replaced: its results are complete); -2: no pool thread */
static int par_start(int n, env_fn fn, void* ctx) {
  pthread_mutex_lock(&P.mu);
  if ((P.async_busy && atomic_load(&P.remaining) > 0) || P.sync_busy) { pthread_mutex_unlock(&P.mu); return -1; }
  post(fn, ctx, n, g_threads < n ? g_threads : n);
  if (P.n_workers == 0) { P.async_busy = 0; pthread_mutex_unlock(&P.mu); return -2; }
  P.async_busy = 1;
  P.async_id = (int)(P.gen & 0x3fffffffu) + 1;
  const int id = P.async_id;
  pthread_mutex_unlock(&P.mu);
  return id;
}
/* returns when job `id` is done (at once when it is not the job in flight: a later job can only have started after it);
 * id 0: whatever is in flight */
static int par_wait(int id) {
  pthread_mutex_lock(&P.mu);
  if (!P.async_busy || (id != 0 && P.async_id != id)) { pthread_mutex_unlock(&P.mu); return 0; }
  while (atomic_load(&P.remaining) > 0) pthread_cond_wait(&P.cv_done, &P.mu);
  P.async_busy = 0;
  pthread_mutex_unlock(&P.mu);
  return 0;
}
void v2xsim_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int v2xsim_max_threads(void) { return g_threads; }

/* Environment.py:94-122 (rl/environment.py _v2v_pathloss) */
static double los(double x, double d_bp, double off) {
  if (x < 1e-300) x = 1e-300;
  if (x <= 3) return 22.7 * log10(3.0) + off;
  if (x < d_bp) return 22.7 * log10(x) + off;
  return 40.0 * log10(x) + 9.45 - 17.3 * log10(V2V_H) - 17.3 * log10(V2V_H) + 2.7 * log10(FC / 5);
}
static double nlos(double da, double db, double d_bp, double off) {
  if (db < 1e-300) db = 1e-300;
  double nj = 2.8 - 0.0024 * db;
  if (nj < 1.84) nj = 1.84;
  return los(da, d_bp, off) + 20 - 12.5 * nj + 10 * nj * log10(db) + 3 * log10(FC / 5);
}
static double v2v_pathloss(double x0, double y0, double x1, double y1) {
  const double d1 = fabs(x0 - x1), d2 = fabs(y0 - y1);
  const double d = hypot(d1, d2) + 0.001;
  const double d_bp = 4 * (V2V_H - 1) * (V2V_H - 1) * FC * 1e9 / 3e8;
  const double off = 41 + 20 * log10(FC / 5);
  if ((d1 < d2 ? d1 : d2) < 7) return los(d, d_bp, off);
  const double a = nlos(d1, d2, d_bp, off), b = nlos(d2, d1, d_bp, off);
  return a < b ? a : b;
}
static double v2i_pathloss(double x, double y) {
  const double dist = hypot(fabs(x - BS_X), fabs(y - BS_Y));
  return 128.1 + 37.6 * log10(sqrt(dist * dist + (V2I_H_BS - V2I_H_MS) * (V2I_H_BS - V2I_H_MS)) / 1000);
}

/* One channel update of every environment (BatchedEnviron._channels_of).
 * u[E][n_u]: the step's uniforms, n_u = 2 * ceil((n + n^2 + 2 n rb + 2 n^2 rb) / 2); Gaussian k of an environment is
 * cos / sin of pair k / 2 (random.gauss order).  Draw order inside a step: V2I shadowing (n), V2V shadowing (n^2),
 * V2I fast fading real (n rb) and imaginary (n rb), V2V fast fading real (n^2 rb) and imaginary (n^2 rb).          */
static void channels_env(int n, int rb, const double* ue, int n_u, const double* ve, const double* pe, double* si, double* sv,
                         double* av, double* ai, double* fv, double* fi, double* g /* [n_u] */) {
  const int n_sh = n + n * n, a = n * rb, b = n * n * rb;
  for (int k = 0; k < n_u; k += 2) {
    const double x2pi = ue[k] * TWOPI;
    const double g2rad = sqrt(-2.0 * log(1.0 - ue[k + 1]));
    g[k] = cos(x2pi) * g2rad;
    g[k + 1] = sin(x2pi) * g2rad;
  }
  for (int i = 0; i < n; ++i) {
    const double dd = 0.002 * ve[i];
    si[i] = exp(-1 * (dd / V2I_DECORR)) * si[i] + sqrt(1 - exp(-2 * (dd / V2I_DECORR))) * (g[i] * V2I_SHADOW_STD);
    ai[i] = v2i_pathloss(pe[2 * i], pe[2 * i + 1]) + si[i];
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const double ddm = 0.002 * ve[i] + 0.002 * ve[j];
      const int ij = i * n + j;
      sv[ij] = exp(-1 * (ddm / V2V_DECORR)) * sv[ij] + sqrt(1 - exp(-2 * (ddm / V2V_DECORR))) * (g[n + ij] * V2V_SHADOW_STD);
      av[ij] = v2v_pathloss(pe[2 * i], pe[2 * i + 1], pe[2 * j], pe[2 * j + 1]) + sv[ij] + (i == j ? 50.0 : 0.0);
    }
  const double* f = g + n_sh;
  const double rs2 = 1 / sqrt(2.0);
  for (int k = 0; k < a; ++k) {                  /* 20 log10 |(re + j im) / sqrt 2| */
    const double re = rs2 * f[k], im = rs2 * f[a + k];
    fi[k] = ai[k / rb] - 20 * log10(hypot(re, im));
  }
  for (int k = 0; k < b; ++k) {
    const double re = rs2 * f[2 * a + k], im = rs2 * f[2 * a + b + k];
    fv[k] = av[k / rb] - 20 * log10(hypot(re, im));
  }
}
typedef struct { int n, rb, n_u; const double *u, *vel, *pos; double *v2i_shadow, *v2v_shadow, *v2v_abs, *v2i_abs, *v2v_ff, *v2i_ff, *scratch; } channels_ctx;
static void channels_one(int e, void* p) {
  const channels_ctx* c = (const channels_ctx*)p;
  const int n = c->n, rb = c->rb, n_u = c->n_u;
  channels_env(n, rb, c->u + (int64_t)e * n_u, n_u, c->vel + (int64_t)e * n, c->pos + (int64_t)e * n * 2, c->v2i_shadow + (int64_t)e * n,
               c->v2v_shadow + (int64_t)e * n * n, c->v2v_abs + (int64_t)e * n * n, c->v2i_abs + (int64_t)e * n,
               c->v2v_ff + (int64_t)e * n * n * rb, c->v2i_ff + (int64_t)e * n * rb, c->scratch + (int64_t)e * n_u);
}
void v2xsim_channels(int E, int n, int rb, const double* u, int n_u, const double* vel, const double* pos,
                     double* v2i_shadow, double* v2v_shadow, double* v2v_abs, double* v2i_abs, double* v2v_ff,
                     double* v2i_ff, double* scratch /* [E][n_u] */) {
  channels_ctx c = { n, rb, n_u, u, vel, pos, v2i_shadow, v2v_shadow, v2v_abs, v2i_abs, v2v_ff, v2i_ff, scratch };
  par_for(E, channels_one, &c);
}

/* compute_reward_with_channel_selection (Environment.py:408-458; every link active, one receiver per link).
 * ch[E][n] chosen resource block, dest[E][n] receiver of link k; out: v2v_rate[E][n], v2i_rate[E][m], m = min(rb, n),
 * interference[E][rb] (without noise), v2i_interf[E][rb] and v2v_interf[E][n] (with noise).                          */
typedef struct { int n, rb; const int64_t *ch, *dest; const double *v2v_ff, *v2i_ff, *v2i_abs; double p_v2v, p_v2i, veh_gain, bs_gain, bs_nf,
                 veh_nf, sig2; double *v2v_rate, *v2i_rate, *interference, *v2i_interf, *v2v_interf; } reward_ctx;
static void reward_one(int e, void* p) {
  const reward_ctx* a = (const reward_ctx*)p;
  const int n = a->n, rb = a->rb, m = rb < n ? rb : n;
  const double p_v2v = a->p_v2v, p_v2i = a->p_v2i, veh_gain = a->veh_gain, bs_gain = a->bs_gain, bs_nf = a->bs_nf, sig2 = a->sig2;
  const double gain = 2 * veh_gain - a->veh_nf;
  const int64_t* c = a->ch + (int64_t)e * n;
  const int64_t* d = a->dest + (int64_t)e * n;
  const double* vv = a->v2v_ff + (int64_t)e * n * n * rb;
  const double* vi = a->v2i_ff + (int64_t)e * n * rb;
  double* itf = a->interference + (int64_t)e * rb;
  double* v2i_interf = a->v2i_interf + (int64_t)e * rb;
  for (int r = 0; r < rb; ++r) itf[r] = 0.0;
  for (int k = 0; k < n; ++k)                    /* (at_bs * onehot).sum(axis=1): ascending k per block */
    itf[c[k]] += pow(10.0, (p_v2v - vi[k * rb + c[k]] + veh_gain + bs_gain - bs_nf) / 10);
  for (int r = 0; r < rb; ++r) v2i_interf[r] = itf[r] + sig2;
  for (int k = 0; k < n; ++k) {
    const int64_t rx = d[k], r = c[k];
    const double signal = pow(10.0, (p_v2v - vv[(k * n + rx) * rb + r] + gain) / 10);
    double acc = 0.0;
    if (r < n) acc += pow(10.0, (p_v2i - vv[(r * n + rx) * rb + r] + gain) / 10);   /* the V2I transmitter of block r is vehicle r */
    double cross = 0.0;
    for (int j = 0; j < n; ++j)
      if (j != k && c[j] == r) cross += pow(10.0, (p_v2v - vv[(j * n + rx) * rb + r] + gain) / 10);
    acc += cross;
    const double tot = acc + sig2;
    a->v2v_interf[(int64_t)e * n + k] = tot;
    a->v2v_rate[(int64_t)e * n + k] = log2(1 + signal / tot);
  }
  for (int k = 0; k < m; ++k) {
    const double s = p_v2i - a->v2i_abs[(int64_t)e * n + k] + veh_gain + bs_gain - bs_nf;
    a->v2i_rate[(int64_t)e * m + k] = log2(1 + pow(10.0, s / 10) / v2i_interf[k]);
  }
}
void v2xsim_reward(int E, int n, int rb, const int64_t* ch, const int64_t* dest, const double* v2v_ff, const double* v2i_ff,
                   const double* v2i_abs, double p_v2v, double p_v2i, double veh_gain, double bs_gain, double bs_nf,
                   double veh_nf, double sig2, double* v2v_rate, double* v2i_rate, double* interference,
                   double* v2i_interf, double* v2v_interf) {
  reward_ctx c = { n, rb, ch, dest, v2v_ff, v2i_ff, v2i_abs, p_v2v, p_v2i, veh_gain, bs_gain, bs_nf, veh_nf, sig2,
                   v2v_rate, v2i_rate, interference, v2i_interf, v2v_interf };
  par_for(E, reward_one, &c);
}

/* Compute_Interference (Environment.py:460-493, observable part): out[E][n][rb] in dB */
static inline void interference_row(int n, int rb, int k, const int64_t* dest, const double* vv, double p_v2i, double veh_gain,
                                    double veh_nf, double sig2, double* out) {
  const int64_t rx = dest[k];
  for (int r = 0; r < rb; ++r) {
    double v = sig2;
    /* numpy indexes vehicle number r as the block's V2I transmitter; r < n is the caller's precondition (rb <= n) */
    v += pow(10.0, (p_v2i - vv[((int64_t)r * n + rx) * rb + r] + 2 * veh_gain - veh_nf) / 10);
    out[(int64_t)k * rb + r] = 10 * log10(v);
  }
}
static void interference_env(int n, int rb, const int64_t* dest, const double* vv, double p_v2i, double veh_gain, double veh_nf,
                             double sig2, double* out) {
  for (int k = 0; k < n; ++k) interference_row(n, rb, k, dest, vv, p_v2i, veh_gain, veh_nf, sig2, out);
}
typedef struct { int n, rb; const int64_t* dest; const double* v2v_ff; double p_v2i, veh_gain, veh_nf, sig2; double* out; } interf_ctx;
static void interference_one(int e, void* p) {
  const interf_ctx* c = (const interf_ctx*)p;
  interference_env(c->n, c->rb, c->dest + (int64_t)e * c->n, c->v2v_ff + (int64_t)e * c->n * c->n * c->rb, c->p_v2i, c->veh_gain,
                   c->veh_nf, c->sig2, c->out + (int64_t)e * c->n * c->rb);
}
void v2xsim_interference(int E, int n, int rb, const int64_t* dest, const double* v2v_ff, double p_v2i, double veh_gain,
                         double veh_nf, double sig2, double* out) {
  interf_ctx c = { n, rb, dest, v2v_ff, p_v2i, veh_gain, veh_nf, sig2, out };
  par_for(E, interference_one, &c);
}

/* Agent.observe for one environment (BS_brain.py:389-407, :441-445, :458-467): state[n][3C+1], adj[n][n]; and, when xe is
 * given, the same observation in the engine's packed form (include/v2xgnn.h, rl/replay.py): xe[n][16] float32 = the state row
 * cast to float32 + zero padding (packing.pack_xe), mask[q] = bit p set when p sends to q, col[n (n-2)] = the CSR sources by
 * destination (ascending) when every link has in-degree n-2 ("regular": no link is its own receiver), zeros otherwise.    */
/* the observation row of link k (state[k][3C+1], and its packed float32 copy xe[k][16] when xe is given) */
static inline void observe_row(int n, int C, int k, const int64_t* d, const double* vv, const double* vi, double power, double* st, float* xe) {
  const double A = 80, Bc = 60;
  const int W = 3 * C + 1;
  const int64_t rx = d[k];
  for (int c = 0; c < C; ++c) {
    const double chv = (vv[(k * n + rx) * C + c] - A) / Bc;
    double tot = 0.0;
    for (int p = 0; p < n; ++p) tot += vv[(p * n + rx) * C + c];          /* np.sum over p, ascending */
    const double edge = (((tot - vv[(rx * n + rx) * C + c]) - (n - 1) * A) / Bc - chv) / (n - 2);
    st[k * W + c] = chv;
    st[k * W + C + c] = (vi[k * C + c] - A) / Bc;
    st[k * W + 2 * C + 1 + c] = edge;
  }
  st[k * W + 2 * C] = power;
  if (xe)
    for (int c = 0; c < 16; ++c) xe[k * 16 + c] = c < W ? (float)st[k * W + c] : 0.0f;
}
static void observe_env(int n, int C, const int64_t* d, const double* vv, const double* vi, double power, double* st, double* ad,
                        float* xe, int32_t* mask, int32_t* col, uint8_t* regular) {
  for (int p = 0; p < n; ++p)
    for (int q = 0; q < n; ++q) ad[p * n + q] = p == q ? 0.0 : 1.0;
  for (int k = 0; k < n; ++k) {
    ad[d[k] * n + k] = 0.0;
    observe_row(n, C, k, d, vv, vi, power, st, xe);
  }
  if (!xe) return;
  int reg = 1;
  for (int k = 0; k < n; ++k)
    if (d[k] == k) reg = 0;
  *regular = (uint8_t)reg;
  int o = 0;
  for (int q = 0; q < n; ++q) {
    uint32_t m = 0;
    for (int p = 0; p < n; ++p)
      if (ad[p * n + q] != 0.0) {
        m |= 1u << p;
        if (reg) col[o++] = p;
      }
    mask[q] = (int32_t)m;
  }
  if (!reg)
    for (int k = 0; k < n * (n - 2); ++k) col[k] = 0;
}
typedef struct { int n, C; const int64_t* dest; const double *v2v_ff, *v2i_ff; double power; double *state, *adj; float* xe;
                 int32_t *mask, *col; uint8_t* regular; } observe_ctx;
static void observe_one(int e, void* p) {
  const observe_ctx* c = (const observe_ctx*)p;
  const int n = c->n, C = c->C, W = 3 * C + 1, ne = n * (n - 2);
  observe_env(n, C, c->dest + (int64_t)e * n, c->v2v_ff + (int64_t)e * n * n * C, c->v2i_ff + (int64_t)e * n * C, c->power,
              c->state + (int64_t)e * n * W, c->adj + (int64_t)e * n * n, c->xe ? c->xe + (int64_t)e * n * 16 : 0,
              c->xe ? c->mask + (int64_t)e * n : 0, c->xe ? c->col + (int64_t)e * ne : 0, c->xe ? c->regular + e : 0);
}
void v2xsim_observe(int E, int n, int C, const int64_t* dest, const double* v2v_ff, const double* v2i_ff, double power,
                    double* state, double* adj) {
  observe_ctx c = { n, C, dest, v2v_ff, v2i_ff, power, state, adj, 0, 0, 0, 0 };
  par_for(E, observe_one, &c);
}
/* ... with the packed form beside it (n <= 31, W <= 16) */
void v2xsim_observe_packed(int E, int n, int C, const int64_t* dest, const double* v2v_ff, const double* v2i_ff, double power,
                           double* state, double* adj, float* xe, int32_t* mask, int32_t* col, uint8_t* regular) {
  observe_ctx c = { n, C, dest, v2v_ff, v2i_ff, power, state, adj, xe, mask, col, regular };
  par_for(E, observe_one, &c);
}


/* ---- MT19937 (the generator behind CPython's random and numpy's legacy RandomState), one state per environment ----
 * key[624] + pos exactly as numpy's RandomState.get_state() reports them; doubles as random.random() /
 * random_sample() builds them: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53.                                             */
static void mt_reload(uint32_t* mt) {
  const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAG = 0x9908b0dfu;
  int kk;
  for (kk = 0; kk < 624 - 397; ++kk) {
    const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO);
    mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
  }
  for (; kk < 623; ++kk) {
    const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO);
    mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
  }
  const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
  mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
}
static inline uint32_t mt_next(uint32_t* mt, int32_t* pos) {
  if (*pos >= 624) { mt_reload(mt); *pos = 0; }
  uint32_t y = mt[(*pos)++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
/* The next `count` doubles of a stream in bulk: the state array is consumed block by block -- tempering and the
 * (a >> 5, b >> 6) -> double conversion are loops over arrays the compiler vectorises -- instead of one call, one position check
 * and one scalar tempering per 32-bit output (26 us per 3780 doubles: the sequential part of a 20-link simulator step).  Same
 * outputs in the same order as `count` calls of mt_double. */
static void mt_fill_doubles(uint32_t* mt, int32_t* pos, double* out, int count) {
  uint32_t tmp[1248];
  int done = 0;
  while (done < count) {
    const int want = count - done < 624 ? count - done : 624;     /* doubles of this round: <= 1248 outputs */
    int got = 0;
    const int need = 2 * want;
    while (got < need) {
      if (*pos >= 624) { mt_reload(mt); *pos = 0; }
      int take = 624 - *pos;
      if (take > need - got) take = need - got;
      const uint32_t* src = mt + *pos;
      for (int i = 0; i < take; ++i) {
        uint32_t y = src[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        tmp[got + i] = y;
      }
      got += take;
      *pos += take;
    }
    for (int i = 0; i < want; ++i) {
      const uint32_t a = tmp[2 * i] >> 5, b = tmp[2 * i + 1] >> 6;
      out[done + i] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
    done += want;
  }
}
/* out[e][0..n_u) = the next n_u doubles of stream e */
typedef struct { uint32_t* keys; int32_t* pos; double* out; int n_u; } uniforms_ctx;
static void uniforms_one(int e, void* q) {
  const uniforms_ctx* c = (const uniforms_ctx*)q;
  uint32_t* mt = c->keys + (int64_t)e * 624;
  int32_t p = c->pos[e];
  double* o = c->out + (int64_t)e * c->n_u;
  mt_fill_doubles(mt, &p, o, c->n_u);
  c->pos[e] = p;
}
void v2xsim_mt_uniforms(int E, uint32_t* keys, int32_t* pos, double* out, int n_u) {
  uniforms_ctx c = { keys, pos, out, n_u };
  par_for(E, uniforms_one, &c);
}

/* ---- the scalar integer draws of an episode reset, on the environments' own MT19937 states ------------------------------
 * random.randrange / randint / sample(population, 1) of CPython 3.10 draw for draw: _randbelow_with_getrandbits(n) takes
 * k = n.bit_length() bits (the top k of one 32-bit output) until the value is < n.                                      */
static inline uint32_t mt_below(uint32_t* mt, int32_t* pos, uint32_t n) {
  const int k = 32 - __builtin_clz(n);
  uint32_t r;
  do { r = mt_next(mt, pos) >> (32 - k); } while (r >= n);
  return r;
}
/* add_new_vehicles_by_number (Environment.py:217-234) for every environment: n / 4 groups of one vehicle per direction
 * (down = 1, up = 0, left = 2, right = 3) on a random lane index; per vehicle the position draw, then the velocity draw. */
typedef struct { int n, n_lanes, width, height; uint32_t* keys; int32_t* pos; const double *down, *up, *left, *right; double* xy;
                 int8_t* dirs; double* vel; } reset_ctx;
static void reset_one(int e, void* q) {
  const reset_ctx* c = (const reset_ctx*)q;
  const int n = c->n, n_lanes = c->n_lanes, width = c->width, height = c->height;
  const double *down = c->down, *up = c->up, *left = c->left, *right = c->right;
  uint32_t* mt = c->keys + (int64_t)e * 624;
  int32_t p = c->pos[e];
  double* x = c->xy + (int64_t)e * n * 2;
  int8_t* d = c->dirs + (int64_t)e * n;
  double* v = c->vel + (int64_t)e * n;
  int k = 0;
  for (int g = 0; g < n / 4; ++g) {
    const uint32_t ind = mt_below(mt, &p, (uint32_t)n_lanes);
    x[2 * k] = down[ind]; x[2 * k + 1] = (double)mt_below(mt, &p, (uint32_t)height + 1); d[k] = 1; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
    x[2 * k] = up[ind];   x[2 * k + 1] = (double)mt_below(mt, &p, (uint32_t)height + 1); d[k] = 0; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
    x[2 * k] = (double)mt_below(mt, &p, (uint32_t)width + 1); x[2 * k + 1] = left[ind];  d[k] = 2; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
    x[2 * k] = (double)mt_below(mt, &p, (uint32_t)width + 1); x[2 * k + 1] = right[ind]; d[k] = 3; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
  }
  c->pos[e] = p;
}
void v2xsim_reset_vehicles(int E, int n, uint32_t* keys, int32_t* pos, int n_lanes, const double* down, const double* up,
                           const double* left, const double* right, int width, int height, double* xy, int8_t* dirs,
                           double* vel) {
  reset_ctx c = { n, n_lanes, width, height, keys, pos, down, up, left, right, xy, dirs, vel };
  par_for(E, reset_one, &c);
}
/* renew_neighbor's destination draw (Environment.py:375): random.sample(candidates, 1)[0] for every link -- a population
 * of m <= 21 entries takes CPython's pool method, whose first pick is pool[_randbelow(m)].                              */
typedef struct { int n, m; uint32_t* keys; int32_t* pos; const int64_t* cand; int64_t* dest; } dest_ctx;
static void dest_one(int e, void* q) {
  const dest_ctx* c = (const dest_ctx*)q;
  uint32_t* mt = c->keys + (int64_t)e * 624;
  int32_t p = c->pos[e];
  for (int i = 0; i < c->n; ++i)
    c->dest[(int64_t)e * c->n + i] = c->cand[((int64_t)e * c->n + i) * c->m + mt_below(mt, &p, (uint32_t)c->m)];
  c->pos[e] = p;
}
void v2xsim_sample_dest(int E, int n, int m, uint32_t* keys, int32_t* pos, const int64_t* cand, int64_t* dest) {
  dest_ctx c = { n, m, keys, pos, cand, dest };
  par_for(E, dest_one, &c);
}

/* ---- one whole simulator step of every environment in one call, and the same step computed AHEAD of the agent --------------
 * Nothing in a simulator step depends on the agent's actions except the rates it is paid (v2xsim_reward, on the channels
 * BEFORE the step): mobility, shadowing, fast fading, the observable interference and the next observation are functions of
 * the environment's own state and random stream (Environment.py:236-406, :460-493; BS_brain.py:366-376 calls them in this
 * order after computing the reward).  v2xsim_advance therefore maps (state in) -> (state out, channels, observation) without
 * touching its inputs, and v2xsim_advance_start / _wait run it on a worker thread while the caller scores the observation on
 * the GPU and replays: rl/batched_env.py commits the result when the agent acts, or drops it (inputs untouched) when
 * anything else happens first.                                                                                            */
static inline double mt_double(uint32_t* mt, int32_t* pos) {
  const uint32_t a = mt_next(mt, pos) >> 5, b = mt_next(mt, pos) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* renew_positions (Environment.py:236-345) of one environment, vehicles in index order: a vehicle that reaches a crossing lane
 * draws one uniform per reached lane (in the reference's checking order) and turns with probability 0.4; vehicles that left
 * the map re-enter on the outermost lane.  dirs: 0 up, 1 down, 2 left, 3 right (rl/batched_env.py _DIRS).               */
static void positions_env(int n, uint32_t* mt, int32_t* mtpos, double* xy, int8_t* dirs, const double* vel, double timestep,
                          int n_lanes, const double* up, const double* down, const double* left, const double* right,
                          double width, double height) {
  for (int v = 0; v < n; ++v) {
    const int d = dirs[v];
    const int ax = d < 2 ? 1 : 0;                          /* moving axis: y for up / down */
    const double sg = (d == 0 || d == 3) ? 1.0 : -1.0;
    const double dv = vel[v] * timestep;
    const double av = xy[2 * v + ax], ov = xy[2 * v + 1 - ax];
    /* crossing lanes in checking order: (left, then right) for vertical movers, (up, then down) for horizontal ones */
    const double* tab[2];
    int new_dir[2];
    double side[2], gs[2];
    if (ax == 1) { tab[0] = left; new_dir[0] = 2; side[0] = -1; gs[0] = -1; tab[1] = right; new_dir[1] = 3; side[1] = 1; gs[1] = 1; }
    else         { tab[0] = up;   new_dir[0] = 0; side[0] = 1;  gs[0] = -1; tab[1] = down;  new_dir[1] = 1; side[1] = -1; gs[1] = -1; }
    int turned = 0;
    for (int o = 0; o < 2 && !turned; ++o)
      for (int l = 0; l < n_lanes; ++l) {
        const double lane = tab[o][l];
        const int reached = sg > 0 ? (av <= lane && av + dv >= lane) : (av >= lane && av - dv <= lane);
        if (reached && mt_double(mt, mtpos) < 0.4) {
          const double gap = sg * (lane - av);
          const double new_o = ov + side[o] * (dv + gs[o] * gap);
          if (ax == 1) { xy[2 * v] = new_o; xy[2 * v + 1] = lane; }
          else         { xy[2 * v] = lane;  xy[2 * v + 1] = new_o; }
          dirs[v] = (int8_t)new_dir[o];
          turned = 1;
          break;
        }
      }
    if (!turned) xy[2 * v + ax] = sg > 0 ? av + dv : av - dv;
  }
  for (int v = 0; v < n; ++v) {
    const double x = xy[2 * v], y = xy[2 * v + 1];
    if (x < 0 || y < 0 || x > width || y > height) {
      switch (dirs[v]) {
        case 0: dirs[v] = 3; xy[2 * v + 1] = right[n_lanes - 1]; break;
        case 1: dirs[v] = 2; xy[2 * v + 1] = left[0]; break;
        case 2: dirs[v] = 0; xy[2 * v] = up[0]; break;
        default: dirs[v] = 1; xy[2 * v] = down[n_lanes - 1]; break;
      }
    }
  }
}
typedef struct { int n, n_lanes; uint32_t* keys; int32_t* pos; double* xy; int8_t* dirs; const double* vel; double timestep;
                 const double *up, *down, *left, *right; double width, height; } positions_ctx;
static void positions_one(int e, void* q) {
  const positions_ctx* c = (const positions_ctx*)q;
  int32_t p = c->pos[e];
  positions_env(c->n, c->keys + (int64_t)e * 624, &p, c->xy + (int64_t)e * c->n * 2, c->dirs + (int64_t)e * c->n, c->vel + (int64_t)e * c->n,
                c->timestep, c->n_lanes, c->up, c->down, c->left, c->right, c->width, c->height);
  c->pos[e] = p;
}
void v2xsim_positions(int E, int n, uint32_t* keys, int32_t* pos, double* xy, int8_t* dirs, const double* vel, double timestep,
                      int n_lanes, const double* up, const double* down, const double* left, const double* right, double width,
                      double height) {
  positions_ctx c = { n, n_lanes, keys, pos, xy, dirs, vel, timestep, up, down, left, right, width, height };
  par_for(E, positions_one, &c);
}


static void advance_one(int e, void* q) {
  const v2xsim_advance_args* a = (const v2xsim_advance_args*)q;
  const int n = a->n, rb = a->rb;
  const int n_draws = n + n * n + 2 * n * rb + 2 * n * n * rb, n_u = (n_draws + 1) & ~1;
  const int W = 3 * rb + 1, ne = n * (n - 2);
  uint32_t* mt = a->keys + (int64_t)e * 624;
  double* xy = a->xy + (int64_t)e * n * 2;
  int8_t* dirs = a->dirs + (int64_t)e * n;
  double* si = a->v2i_shadow + (int64_t)e * n;
  double* sv = a->v2v_shadow + (int64_t)e * n * n;
  if (a->keys != a->keys_in) memcpy(mt, a->keys_in + (int64_t)e * 624, 624 * sizeof(uint32_t));
  if (a->xy != a->xy_in) memcpy(xy, a->xy_in + (int64_t)e * n * 2, (size_t)n * 2 * sizeof(double));
  if (a->dirs != a->dirs_in) memcpy(dirs, a->dirs_in + (int64_t)e * n, (size_t)n);
  if (a->v2i_shadow != a->v2i_shadow_in) memcpy(si, a->v2i_shadow_in + (int64_t)e * n, (size_t)n * sizeof(double));
  if (a->v2v_shadow != a->v2v_shadow_in) memcpy(sv, a->v2v_shadow_in + (int64_t)e * n * n, (size_t)n * n * sizeof(double));
  int32_t p = a->mtpos_in[e];
  const double* ve = a->vel + (int64_t)e * n;
  positions_env(n, mt, &p, xy, dirs, ve, a->timestep, a->n_lanes, a->up, a->down, a->left, a->right, a->width, a->height);
  double* u = a->scratch + (int64_t)e * 2 * n_u;
  mt_fill_doubles(mt, &p, u, n_u);
  a->mtpos[e] = p;
  double* fv = a->v2v_ff + (int64_t)e * n * n * rb;
  double* fi = a->v2i_ff + (int64_t)e * n * rb;
  channels_env(n, rb, u, n_u, ve, xy, si, sv, a->v2v_abs + (int64_t)e * n * n, a->v2i_abs + (int64_t)e * n, fv, fi, u + n_u);
  const int64_t* d = a->dest + (int64_t)e * n;
  interference_env(n, rb, d, fv, a->p_v2i, a->veh_gain, a->veh_nf, a->sig2, a->interf_db + (int64_t)e * n * rb);
  observe_env(n, rb, d, fv, fi, a->p_v2v, a->state + (int64_t)e * n * W, a->adj + (int64_t)e * n * n, a->xe + (int64_t)e * n * 16,
              a->mask + (int64_t)e * n, a->col + (int64_t)e * ne, a->regular + e);
}
void v2xsim_advance(const v2xsim_advance_args* a) {
  v2xsim_advance_args c = *a;
  par_for(c.E, advance_one, &c);
}

/* the same step on the pool alone while the caller does something else: one job in flight per process */
static v2xsim_advance_args g_job;
int v2xsim_advance_start(const v2xsim_advance_args* a) {
  pthread_mutex_lock(&P.mu);
  const int running = P.async_busy && atomic_load(&P.remaining) > 0;
  pthread_mutex_unlock(&P.mu);
  if (running) return -1;
  g_job = *a;
  return par_start(g_job.E, advance_one, &g_job);
}
int v2xsim_advance_wait(int id) { return par_wait(id); }

/* ---- the reference's own rollout shape as ONE call: one simulator, T sequential transitions ------------------------------------
 * Agent.generate_d2d_transition (BS_brain.py:409-553): observe -> epsilon-greedy action (a B = 1 predict when greedy, :308-352)
 * -> rates on the current channels -> simulator step -> next observation, T = 50 times before every replay (:818-832).  One
 * simulator step of 20 links is ~200 us of libm on one thread (3780 Gaussians, 400 path losses, 1680 fast-fading terms), so
 * the loop the reference runs is bound by the simulator, not by the predict: here a step is cut over a small team of threads
 * that spin for the duration of the call (they sleep between calls: cgroup CPU quota, see the pool above):
 *   stream thread   step k: mobility with its turn draws, then the step's n_u uniforms of the environment's MT19937 stream --
 *                   the only inherently sequential part (~30 us); depends on nothing but its own previous step, so it runs up to
 *                   RO_RING - 1 steps ahead into a ring
 *   K workers       step k: Gaussians (pairs split), barrier, shadowing / path loss / fast fading (link rows split), barrier,
 *                   observable interference + observation (worker 0)
 *   caller          transition k: epsilon draw and random actions on numpy's process-wide MT19937 (draw for draw: random_sample,
 *                   randint's masked rejection), or the predict through a callback + first-maximiser argmax; rates on the channels of
 *                   state k (v2xsim_reward's arithmetic); then waits for state k + 1
 * The workers compute state k + 1 while the caller is inside transition k (nothing in a step depends on the actions), into
 * the buffer of state k - 1, which the caller has left.  Every element is computed by the same expressions as
 * v2xsim_advance / v2xsim_reward: states, rates and both random streams are bit-identical to T single steps
 * (tests/test_rl_batched_env.py).  g_threads <= 2: everything on the caller, same results.                                 */
static inline uint32_t np_interval(uint32_t* mt, int32_t* pos, uint32_t max);   /* below: numpy's masked rejection */
#define RO_RING 4
#define RO_MAXW 12
typedef struct {
  double *v2v_abs, *v2i_abs, *v2v_ff, *v2i_ff, *interf_db, *state, *adj;
  float* xe; int32_t *mask, *col; uint8_t regular;
} ro_state;
typedef struct { double* xy; int8_t* dirs; double *u, *g; } ro_slot;
typedef struct {
  const v2xsim_rollout_args* a;
  int n, rb, n_u, n_sh, T, K;
  ro_state* st; int ns;                           /* state s lives in st[s % ns]: ns = 2 (a state's buffer is re-used two steps
                                                     later), or T + 1 when every state of the call is kept (batch_predict) */
  ro_slot ring[RO_RING];
  uint32_t mt[624]; int32_t mtpos;                 /* the environment's stream (owned by the stream thread during the call) */
  double *si, *sv;                                 /* shadowing, updated in place by the workers */
  double* gw[RO_MAXW];                             /* batch mode: a private Gaussian scratch per worker [n_u] */
  int batch;
  const ro_state* s0;                              /* state 0 (the caller's arrays), for the team's record phase */
  _Atomic int records_go, records_done;            /* batch mode: n_ok + 1 once the actions are known; workers that finished */
  _Atomic int stream_done, main_pos, workers_done, bar, stop;
  _Atomic int steps_of[RO_MAXW];                   /* batch mode: steps worker w has finished (the stream thread's ring follows the slowest) */
} ro_ctx;

static void ro_spin(_Atomic int* v, int want) {    /* until *v >= want */
  while (atomic_load_explicit(v, memory_order_acquire) < want) __builtin_ia32_pause();
}
/* stream part of step k: positions of state k + 1 and the step's uniforms */
static void ro_stream_step(ro_ctx* c, int k) {
  const v2xsim_rollout_args* a = c->a;
  const int n = c->n;
  ro_slot* s = &c->ring[k % RO_RING];
  const double* xy_prev = k == 0 ? a->xy : c->ring[(k - 1) % RO_RING].xy;
  const int8_t* d_prev = k == 0 ? a->dirs : c->ring[(k - 1) % RO_RING].dirs;
  memcpy(s->xy, xy_prev, (size_t)n * 2 * sizeof(double));
  memcpy(s->dirs, d_prev, (size_t)n);
  positions_env(n, c->mt, &c->mtpos, s->xy, s->dirs, a->vel, a->timestep, a->n_lanes, a->up, a->down, a->left, a->right, a->width, a->height);
  mt_fill_doubles(c->mt, &c->mtpos, s->u, c->n_u);
}
/* worker `w` of `K`, the three phases of step k (channels_env / interference_env / observe_env cut by index ranges) */
static void ro_gauss(ro_ctx* c, int k, int w, int K) {
  const ro_slot* s = &c->ring[k % RO_RING];
  const int pairs = c->n_u / 2, p0 = (int)((int64_t)pairs * w / K), p1 = (int)((int64_t)pairs * (w + 1) / K);
  for (int q = p0; q < p1; ++q) {
    const double x2pi = s->u[2 * q] * TWOPI;
    const double g2rad = sqrt(-2.0 * log(1.0 - s->u[2 * q + 1]));
    s->g[2 * q] = cos(x2pi) * g2rad;
    s->g[2 * q + 1] = sin(x2pi) * g2rad;
  }
}
static void ro_channels(ro_ctx* c, int k, int w, int K) {
  const v2xsim_rollout_args* a = c->a;
  const int n = c->n, rb = c->rb, n_sh = c->n_sh, na = n * rb, nb = n * n * rb;
  const ro_slot* s = &c->ring[k % RO_RING];
  ro_state* o = &c->st[(k + 1) % c->ns];
  const double *g = s->g, *pe = s->xy, *ve = a->vel;
  const double* f = g + n_sh;
  const double rs2 = 1 / sqrt(2.0);
  const int i0 = (int)((int64_t)n * w / K), i1 = (int)((int64_t)n * (w + 1) / K);
  for (int i = i0; i < i1; ++i) {                  /* V2I: shadowing, path loss, fast fading of link i */
    const double dd = 0.002 * ve[i];
    c->si[i] = exp(-1 * (dd / V2I_DECORR)) * c->si[i] + sqrt(1 - exp(-2 * (dd / V2I_DECORR))) * (g[i] * V2I_SHADOW_STD);
    o->v2i_abs[i] = v2i_pathloss(pe[2 * i], pe[2 * i + 1]) + c->si[i];
    for (int r = 0; r < rb; ++r) {
      const int q = i * rb + r;
      const double re = rs2 * f[q], im = rs2 * f[na + q];
      o->v2i_ff[q] = o->v2i_abs[i] - 20 * log10(hypot(re, im));
    }
  }
  for (int i = i0; i < i1; ++i)                    /* V2V rows i */
    for (int j = 0; j < n; ++j) {
      const double ddm = 0.002 * ve[i] + 0.002 * ve[j];
      const int ij = i * n + j;
      c->sv[ij] = exp(-1 * (ddm / V2V_DECORR)) * c->sv[ij] + sqrt(1 - exp(-2 * (ddm / V2V_DECORR))) * (g[n + ij] * V2V_SHADOW_STD);
      o->v2v_abs[ij] = v2v_pathloss(pe[2 * i], pe[2 * i + 1], pe[2 * j], pe[2 * j + 1]) + c->sv[ij] + (i == j ? 50.0 : 0.0);
      for (int r = 0; r < rb; ++r) {
        const int q = ij * rb + r;
        const double re = rs2 * f[2 * na + q], im = rs2 * f[2 * na + nb + q];
        o->v2v_ff[q] = o->v2v_abs[ij] - 20 * log10(hypot(re, im));
      }
    }
}
/* observable interference + observation rows of worker w's links.  The parts of an observation that depend on the receivers
 * only -- adjacency, masks, CSR sources, the regular flag -- do not change inside a call: both state buffers carry the copies
 * of the caller's (v2xsim_rollout) */
static void ro_observe(ro_ctx* c, int k, int w, int K) {
  const v2xsim_rollout_args* a = c->a;
  ro_state* o = &c->st[(k + 1) % c->ns];
  const int n = c->n, i0 = (int)((int64_t)n * w / K), i1 = (int)((int64_t)n * (w + 1) / K);
  for (int i = i0; i < i1; ++i) {
    interference_row(n, c->rb, i, a->dest, o->v2v_ff, a->p_v2i, a->veh_gain, a->veh_nf, a->sig2, o->interf_db);
    observe_row(n, c->rb, i, a->dest, o->v2v_ff, o->v2i_ff, a->p_v2v, o->state, o->xe);
  }
}
/* batch mode (every state of the call has its own buffer): a worker owns a FIXED range of the V2V link pairs and of the V2I
 * links through all T steps -- the Gaussians its elements need (computed into a scratch of its own: pairs that straddle a range
 * boundary are simply computed by both neighbours), their shadowing recursion, path loss and fast fading -- and needs nobody
 * but the stream thread until the very end: no barrier per step (three per step made the call as slow as its most-descheduled
 * worker on a shared host).  Same expressions per element as ro_gauss / ro_channels.                                          */
static void ro_gauss_range(const double* u, double* g, int lo, int hi) {          /* Gaussians lo .. hi-1 (whole pairs around them) */
  for (int q = lo >> 1; 2 * q < hi; ++q) {
    const double x2pi = u[2 * q] * TWOPI;
    const double g2rad = sqrt(-2.0 * log(1.0 - u[2 * q + 1]));
    g[2 * q] = cos(x2pi) * g2rad;
    g[2 * q + 1] = sin(x2pi) * g2rad;
  }
}
static void ro_step_owned(ro_ctx* c, int k, int w, int K) {
  const v2xsim_rollout_args* a = c->a;
  const int n = c->n, rb = c->rb, n_sh = c->n_sh, na = n * rb, nb = n * n * rb;
  const ro_slot* s = &c->ring[k % RO_RING];
  ro_state* o = &c->st[(k + 1) % c->ns];
  double* g = c->gw[w];
  const double *pe = s->xy, *ve = a->vel;
  const double rs2 = 1 / sqrt(2.0);
  const int i0 = (int)((int64_t)n * w / K), i1 = (int)((int64_t)n * (w + 1) / K);              /* V2I links */
  const int p0 = (int)((int64_t)n * n * w / K), p1 = (int)((int64_t)n * n * (w + 1) / K);      /* V2V pairs ij */
  ro_gauss_range(s->u, g, i0, i1);
  ro_gauss_range(s->u, g, n + p0, n + p1);
  ro_gauss_range(s->u, g, n_sh + i0 * rb, n_sh + i1 * rb);
  ro_gauss_range(s->u, g, n_sh + na + i0 * rb, n_sh + na + i1 * rb);
  ro_gauss_range(s->u, g, n_sh + 2 * na + p0 * rb, n_sh + 2 * na + p1 * rb);
  ro_gauss_range(s->u, g, n_sh + 2 * na + nb + p0 * rb, n_sh + 2 * na + nb + p1 * rb);
  const double* f = g + n_sh;
  for (int i = i0; i < i1; ++i) {
    const double dd = 0.002 * ve[i];
    c->si[i] = exp(-1 * (dd / V2I_DECORR)) * c->si[i] + sqrt(1 - exp(-2 * (dd / V2I_DECORR))) * (g[i] * V2I_SHADOW_STD);
    o->v2i_abs[i] = v2i_pathloss(pe[2 * i], pe[2 * i + 1]) + c->si[i];
    for (int r = 0; r < rb; ++r) {
      const int q = i * rb + r;
      const double re = rs2 * f[q], im = rs2 * f[na + q];
      o->v2i_ff[q] = o->v2i_abs[i] - 20 * log10(hypot(re, im));
    }
  }
  for (int ij = p0; ij < p1; ++ij) {
    const int i = ij / n, j = ij - i * n;
    const double ddm = 0.002 * ve[i] + 0.002 * ve[j];
    c->sv[ij] = exp(-1 * (ddm / V2V_DECORR)) * c->sv[ij] + sqrt(1 - exp(-2 * (ddm / V2V_DECORR))) * (g[n + ij] * V2V_SHADOW_STD);
    o->v2v_abs[ij] = v2v_pathloss(pe[2 * i], pe[2 * i + 1], pe[2 * j], pe[2 * j + 1]) + c->sv[ij] + (i == j ? 50.0 : 0.0);
    for (int r = 0; r < rb; ++r) {
      const int q = ij * rb + r;
      const double re = rs2 * f[2 * na + q], im = rs2 * f[2 * na + nb + q];
      o->v2v_ff[q] = o->v2v_abs[ij] - 20 * log10(hypot(re, im));
    }
  }
}
static void ro_barrier(ro_ctx* c, int* phase) {    /* among the K workers */
  ++*phase;
  atomic_fetch_add_explicit(&c->bar, 1, memory_order_acq_rel);
  ro_spin(&c->bar, *phase * c->K);
}
static void ro_record(const v2xsim_rollout_args* a, int t, const ro_state* cur, const int64_t* act, double* side);
static void ro_worker(ro_ctx* c, int w) {
  int phase = 0;
  if (c->batch) {                                  /* all steps on the own ranges, ONE barrier, then the observation rows of all states */
    for (int k = 0; k < c->T; ++k) {
      ro_spin(&c->stream_done, k + 1);
      ro_step_owned(c, k, w, c->K);
      atomic_store_explicit(&c->steps_of[w], k + 1, memory_order_release);
    }
    ro_barrier(c, &phase);
    for (int k = 0; k < c->T; ++k) ro_observe(c, k, w, c->K);
    ro_barrier(c, &phase);
    if (w == 0) atomic_store_explicit(&c->workers_done, c->T, memory_order_release);
    /* the caller takes the predict and the argmax; then every worker the rates + records of its transitions */
    while (atomic_load_explicit(&c->records_go, memory_order_acquire) == 0) __builtin_ia32_pause();
    const int n_ok = atomic_load_explicit(&c->records_go, memory_order_acquire) - 1;
    const v2xsim_rollout_args* a = c->a;
    for (int t = w; t < n_ok; t += c->K) {
      const ro_state* cur = t == 0 ? c->s0 : &c->st[t % c->ns];
      ro_record(a, t, cur, a->t_action + (int64_t)t * c->n, t == n_ok - 1 ? 0 : c->gw[w]);
      memcpy(a->t_xe_next + (int64_t)t * c->n * 16, c->st[(t + 1) % c->ns].xe, (size_t)c->n * 16 * sizeof(float));
    }
    atomic_fetch_add_explicit(&c->records_done, 1, memory_order_acq_rel);
    return;
  }
  for (int k = 0; k < c->T; ++k) {
    ro_spin(&c->stream_done, k + 1);               /* the step's positions and uniforms */
    ro_spin(&c->main_pos, k);                      /* the caller has left state k - 1, whose buffer state k + 1 takes */
    ro_gauss(c, k, w, c->K);
    ro_barrier(c, &phase);
    ro_channels(c, k, w, c->K);
    ro_barrier(c, &phase);
    ro_observe(c, k, w, c->K);
    ro_barrier(c, &phase);
    if (w == 0) atomic_store_explicit(&c->workers_done, k + 1, memory_order_release);
  }
}
static void ro_stream(ro_ctx* c) {
  for (int k = 0; k < c->T; ++k) {
    if (c->batch) {                                /* the ring slot's previous tenant: consumed by EVERY worker */
      for (int w = 0; w < c->K; ++w) ro_spin(&c->steps_of[w], k - (RO_RING - 1));
    } else
    ro_spin(&c->workers_done, k - (RO_RING - 1));  /* the ring slot's previous tenant (step k - RO_RING) has been consumed */
    ro_stream_step(c, k);
    atomic_store_explicit(&c->stream_done, k + 1, memory_order_release);
  }
}
/* the team: threads that live across calls and sleep between them */
static struct {
  pthread_mutex_t mu; pthread_cond_t cv_go, cv_idle;
  pthread_t th[RO_MAXW + 1]; int n_started;
  ro_ctx* job; uint32_t gen; int n_members, n_left;
  pthread_mutex_t call_mu;                         /* one rollout at a time */
} RT = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, 0, PTHREAD_MUTEX_INITIALIZER };
static void* ro_team_main(void* arg) {
  const int id = (int)(intptr_t)arg;               /* 0: the stream thread, 1 .. K: worker id - 1 */
  uint32_t seen = 0;
  pthread_mutex_lock(&RT.mu);
  for (;;) {
    while (RT.gen == seen) pthread_cond_wait(&RT.cv_go, &RT.mu);
    seen = RT.gen;
    ro_ctx* c = RT.job;
    const int member = id < RT.n_members;
    pthread_mutex_unlock(&RT.mu);
    if (member) {
      if (id == 0) ro_stream(c); else ro_worker(c, id - 1);
    }
    pthread_mutex_lock(&RT.mu);
    if (member && --RT.n_left == 0) pthread_cond_broadcast(&RT.cv_idle);
  }
  return 0;
}
static void ro_after_fork(void) {
  pthread_mutex_init(&RT.mu, 0); pthread_cond_init(&RT.cv_go, 0); pthread_cond_init(&RT.cv_idle, 0);
  pthread_mutex_init(&RT.call_mu, 0);
  RT.n_started = 0; RT.n_left = 0; RT.job = 0;
}
static int ro_team_start(ro_ctx* c, int members) {  /* -> members actually running (0: no thread could be started) */
  static int atfork_set = 0;
  pthread_mutex_lock(&RT.mu);
  if (!atfork_set) { pthread_atfork(0, 0, ro_after_fork); atfork_set = 1; }
  while (RT.n_started < members) {
    if (pthread_create(&RT.th[RT.n_started], 0, ro_team_main, (void*)(intptr_t)RT.n_started) != 0) break;
    pthread_detach(RT.th[RT.n_started]);
    ++RT.n_started;
  }
  if (RT.n_started < members) { pthread_mutex_unlock(&RT.mu); return 0; }
  RT.job = c; RT.n_members = members; RT.n_left = members;
  ++RT.gen;
  pthread_cond_broadcast(&RT.cv_go);
  pthread_mutex_unlock(&RT.mu);
  return members;
}
static void ro_team_join(void) {
  pthread_mutex_lock(&RT.mu);
  while (RT.n_left > 0) pthread_cond_wait(&RT.cv_idle, &RT.mu);
  RT.job = 0;
  pthread_mutex_unlock(&RT.mu);
}

/* numpy's legacy RandomState.randint(0, hi) for hi <= 2^32: masked rejection on one 32-bit output per try
 * (_rand_int64 -> random_bounded_uint64_fill -> buffered_bounded_masked_uint32, numpy/random/src/distributions/distributions.c) */
static inline int64_t np_randint(uint32_t* mt, int32_t* pos, uint32_t hi) {
  const uint32_t rng = hi - 1;
  if (rng == 0) return 0;
  return (int64_t)np_interval(mt, pos, rng);
}

/* The epsilon-greedy draws of ONE iteration over E simulators (Agent._packed_iteration / _generate_batched, BS_brain.py:308-352 per
 * simulator in order): for e = 0 .. E-1 the schedule's epsilon at step step_no0 + e, one random_sample(); below epsilon n
 * randint(0, n_actions) draws -> actions[e][0..n), greedy[e] = 0; else greedy[e] = 1 (its actions come from the predict).  On
 * numpy's process-wide MT19937 (np_key / np_pos: get_state / set_state around the call), draw for draw what the Python loop takes:
 * 50 simulators are ~40 calls of np.random.randint at 4.6 us each early in a run.  Returns the number of greedy simulators. */
int v2xsim_np_policy_draws(uint32_t* np_key, int32_t* np_pos, int32_t E, int32_t n, int32_t n_actions, double eps_max, double eps_min,
                           double eps_per_step, double eps_steps, int64_t step_no0, int64_t* actions, uint8_t* greedy, double* eps_last) {
  if (!np_key || !np_pos || !actions || !greedy || E < 0 || n < 1 || n_actions < 1) return -1;
  int n_greedy = 0;
  double eps = eps_min;
  for (int e = 0; e < E; ++e) {
    const int64_t step_no = step_no0 + e;
    eps = (double)step_no < eps_steps ? eps_max - eps_per_step * (double)step_no : eps_min;
    if (mt_double(np_key, np_pos) < eps) {
      for (int i = 0; i < n; ++i) actions[(int64_t)e * n + i] = np_randint(np_key, np_pos, (uint32_t)n_actions);
      greedy[e] = 0;
    } else {
      greedy[e] = 1;
      ++n_greedy;
    }
  }
  if (eps_last) *eps_last = eps;
  return n_greedy;
}

static double* ro_buf; static size_t ro_buf_len;     /* the call's working memory, kept between calls */
static ro_state* ro_sts; static int ro_sts_len;

/* one transition's policy draws on numpy's generator -> 1 when it explores (act filled), 0 when it is greedy */
static int ro_policy_draw(const v2xsim_rollout_args* a, int t, int64_t* act, double* eps_out) {
  const int64_t step_no = a->step_no0 + t;
  const double eps = (double)step_no < a->eps_steps ? a->eps_max - a->eps_per_step * (double)step_no : a->eps_min;
  *eps_out = eps;
  if (mt_double(a->np_key, a->np_pos) < eps) {
    for (int i = 0; i < a->n; ++i) act[i] = np_randint(a->np_key, a->np_pos, (uint32_t)a->n_actions);
    return 1;
  }
  return 0;
}
static void ro_argmax(const v2xsim_rollout_args* a, const float* q, int64_t* act) {       /* first maximiser (np.argmax) */
  for (int i = 0; i < a->n; ++i) {
    const float* qi = q + (int64_t)i * a->n_actions;
    int best = 0;
    for (int ch = 1; ch < a->n_actions; ++ch) if (qi[ch] > qi[best]) best = ch;
    act[i] = best;
  }
}
/* rates on the channels of state t + the transition's record (everything but xe_next).  side: [2 rb + n] doubles for
 * v2xsim_reward's side outputs (interference, with noise, per link), or null = the caller's arrays (the LAST transition's are kept) */
static void ro_record(const v2xsim_rollout_args* a, int t, const ro_state* cur, const int64_t* act, double* side) {
  const int n = a->n, rb = a->rb, ne = n * (n - 2), m = rb < n ? rb : n;
  reward_ctx r = { n, rb, act, a->dest, cur->v2v_ff, cur->v2i_ff, cur->v2i_abs, a->p_v2v, a->p_v2i, a->veh_gain, a->bs_gain, a->bs_nf,
                   a->veh_nf, a->sig2, a->t_v2v_rate + (int64_t)t * n, a->t_v2i_rate + (int64_t)t * m, side ? side : a->interference,
                   side ? side + rb : a->v2i_interf, side ? side + 2 * rb : a->v2v_interf };
  reward_one(0, &r);
  memcpy(a->t_xe + (int64_t)t * n * 16, cur->xe, (size_t)n * 16 * sizeof(float));
  memcpy(a->t_col + (int64_t)t * ne, cur->col, (size_t)ne * sizeof(int32_t));
  memcpy(a->t_mask + (int64_t)t * n, cur->mask, (size_t)n * sizeof(int32_t));
  a->t_regular[t] = cur->regular;
}

int v2xsim_rollout(v2xsim_rollout_args* a) {
  if (!a || a->n < 3 || a->n > 31 || a->rb < 1 || a->rb > a->n || 3 * a->rb + 1 > 16 || a->T < 1 || a->n_actions < 1) return -1;
  const int n = a->n, rb = a->rb, T = a->T, W = 3 * rb + 1, ne = n * (n - 2);
  const int n_draws = n + n * n + 2 * n * rb + 2 * n * n * rb;
  if (n_draws & 1) return -1;
  if (pthread_mutex_trylock(&RT.call_mu) != 0) return -3;
  const int batch = a->batch_predict != 0;
  const int ns = batch ? T + 1 : 2;
  /* working memory: the state buffers + the ring, carved out of one block */
  const size_t per_state = (size_t)n * n + n + (size_t)n * n * rb + (size_t)n * rb + (size_t)n * rb + (size_t)n * W + (size_t)n * n   /* doubles */
                           + ((size_t)n * 16 * 4 + (size_t)n * 4 + (size_t)(ne > 0 ? ne : 1) * 4 + 7) / 8 + 4;
  const size_t per_slot = (size_t)n * 2 + ((size_t)n + 7) / 8 + 2 * (size_t)n_draws + 2;
  const size_t need = (size_t)ns * per_state + RO_RING * per_slot + (size_t)n + (size_t)n * n + 16
                      + (batch ? ((size_t)T + 7) / 8 + 80 + (size_t)RO_MAXW * n_draws : 0);
  if (ro_buf_len < need) {
    free(ro_buf);
    ro_buf = (double*)malloc(need * sizeof(double));
    ro_buf_len = ro_buf ? need : 0;
    if (!ro_buf) { pthread_mutex_unlock(&RT.call_mu); return -2; }
  }
  if (ro_sts_len < ns) {
    free(ro_sts);
    ro_sts = (ro_state*)malloc((size_t)ns * sizeof(ro_state));
    ro_sts_len = ro_sts ? ns : 0;
    if (!ro_sts) { pthread_mutex_unlock(&RT.call_mu); return -2; }
  }
  ro_ctx c;
  memset(&c, 0, sizeof(c));
  c.a = a; c.n = n; c.rb = rb; c.n_u = n_draws; c.n_sh = n + n * n; c.T = T; c.st = ro_sts; c.ns = ns;
  double* p = ro_buf;
  for (int b = 0; b < ns; ++b) {
    ro_state* s = &c.st[b];
    s->v2v_abs = p; p += (size_t)n * n;
    s->v2i_abs = p; p += n;
    s->v2v_ff = p; p += (size_t)n * n * rb;
    s->v2i_ff = p; p += (size_t)n * rb;
    s->interf_db = p; p += (size_t)n * rb;
    s->state = p; p += (size_t)n * W;
    s->adj = p; p += (size_t)n * n;
    s->xe = (float*)p; p += ((size_t)n * 16 * 4 + 7) / 8;
    s->mask = (int32_t*)p; p += ((size_t)n * 4 + 7) / 8;
    s->col = (int32_t*)p; p += ((size_t)(ne > 0 ? ne : 1) * 4 + 7) / 8;
    memcpy(s->adj, a->adj, (size_t)n * n * sizeof(double));         /* what depends on the receivers only: constant over the call */
    memcpy(s->mask, a->mask, (size_t)n * sizeof(int32_t));
    memcpy(s->col, a->col, (size_t)ne * sizeof(int32_t));
    s->regular = *a->regular;
  }
  for (int r = 0; r < RO_RING; ++r) {
    ro_slot* s = &c.ring[r];
    s->xy = p; p += (size_t)n * 2;
    s->dirs = (int8_t*)p; p += ((size_t)n + 7) / 8;
    s->u = p; p += n_draws;
    s->g = p; p += n_draws;
  }
  c.si = p; p += n;
  c.sv = p; p += (size_t)n * n;
  if (batch)
    for (int w = 0; w < RO_MAXW; ++w) { c.gw[w] = p; p += n_draws; }
  c.batch = batch;
  uint8_t* explores = (uint8_t*)p;                 /* batch_predict: per transition, 1 = random actions drawn */
  /* state 0 = the environment as it stands (channels read in place, never written: buffer 0 is first written as state 2 / never) */
  ro_state s0 = { a->v2v_abs, a->v2i_abs, a->v2v_ff, a->v2i_ff, a->interf_db, a->state, a->adj, a->xe, a->mask, a->col, *a->regular };
  memcpy(c.mt, a->keys, sizeof(c.mt));
  c.mtpos = *a->mtpos;
  memcpy(c.si, a->v2i_shadow, (size_t)n * sizeof(double));
  memcpy(c.sv, a->v2v_shadow, (size_t)n * n * sizeof(double));
  int K = g_threads - 2;
  if (K > RO_MAXW) K = RO_MAXW;
  if (K > n) K = n;
  c.K = K > 0 ? K : 1;
  if (batch) atomic_store(&c.main_pos, T);        /* every state has a buffer of its own: the team never waits for the caller */
  const int team = K >= 1 ? ro_team_start(&c, 1 + c.K) : 0;
  if (!team) c.K = 1;
  int rc = T;
  double eps = a->eps_min;
#define RO_STATE(t_) ((t_) == 0 ? (const ro_state*)&s0 : (const ro_state*)&c.st[(t_) % ns])
  if (batch) {
    /* ---- all policy draws first (they need no Q-value: a greedy transition draws nothing but its epsilon), the states meanwhile
     * on the team; then ONE predict of the T observations (the network does not change inside a rollout and a graph's Q-values do
     * not depend on the batch around it), then argmax / rates / records transition by transition */
    uint32_t np_key0[624];
    memcpy(np_key0, a->np_key, sizeof(np_key0));
    const int32_t np_pos0 = *a->np_pos;
    int first_greedy = -1;
    for (int t = 0; t < T; ++t) {
      explores[t] = (uint8_t)ro_policy_draw(a, t, a->t_action + (int64_t)t * n, &eps);
      if (!explores[t]) { ++*a->n_greedy; if (first_greedy < 0) first_greedy = t; }
    }
    if (!team)
      for (int t = 0; t < T; ++t) { ro_stream_step(&c, t); ro_step_owned(&c, t, 0, 1); ro_observe(&c, t, 0, 1); }
    else
      ro_spin(&c.workers_done, T);
    int n_ok = T;                                  /* transitions that can be completed */
    if (first_greedy >= 0) {
      if (!*a->regular || !a->predict) { rc = -4 - first_greedy; n_ok = first_greedy; }
      else {
        for (int t = 0; t < T; ++t) {
          memcpy(a->xe_pin + (int64_t)t * n * 16, RO_STATE(t)->xe, (size_t)n * 16 * sizeof(float));
          memcpy(a->col_pin + (int64_t)t * ne, RO_STATE(t)->col, (size_t)ne * sizeof(int32_t));
        }
        if (a->predict(a->predict_ctx) != 0) { rc = -1000 - first_greedy; n_ok = first_greedy; }
      }
    }
    for (int t = 0; t < n_ok; ++t) {
      int64_t* act = a->t_action + (int64_t)t * n;
      if (!explores[t]) ro_argmax(a, a->q_pin + (int64_t)t * n * a->n_actions, act);
      if (!team) {
        ro_record(a, t, RO_STATE(t), act, 0);
        memcpy(a->t_xe_next + (int64_t)t * n * 16, c.st[(t + 1) % ns].xe, (size_t)n * 16 * sizeof(float));
      }
    }
    if (team) {                                    /* rates + records of the transitions on the team (t = w, w + K, ...): ~6 us of pow / log2 each */
      c.s0 = &s0;
      atomic_store_explicit(&c.records_go, n_ok + 1, memory_order_release);
      ro_spin(&c.records_done, c.K);
    }
    if (n_ok < T) {                                /* numpy's stream: where the failed transition's epsilon draw left it */
      memcpy(a->np_key, np_key0, sizeof(np_key0));
      *a->np_pos = np_pos0;
      *a->n_greedy = 0;
      for (int t = 0; t <= n_ok; ++t)
        if (!ro_policy_draw(a, t, a->t_action + (int64_t)t * n, &eps)) ++*a->n_greedy;
    }
  } else {
    for (int t = 0; t < T; ++t) {
      const ro_state* cur = RO_STATE(t);
      atomic_store_explicit(&c.main_pos, t, memory_order_release);
      if (!team) {                                 /* no helpers: the step right here */
        ro_stream_step(&c, t);
        ro_gauss(&c, t, 0, 1); ro_channels(&c, t, 0, 1); ro_observe(&c, t, 0, 1);
        atomic_store(&c.stream_done, t + 1); atomic_store(&c.workers_done, t + 1);
      }
      /* ---- epsilon-greedy action (BS_brain.py:308-352) */
      int64_t* act = a->t_action + (int64_t)t * n;
      if (!ro_policy_draw(a, t, act, &eps)) {
        if (!cur->regular || !a->predict) { rc = -4 - t; break; }        /* a link that is its own receiver: the caller's general path */
        memcpy(a->xe_pin, cur->xe, (size_t)n * 16 * sizeof(float));
        memcpy(a->col_pin, cur->col, (size_t)ne * sizeof(int32_t));
        if (a->predict(a->predict_ctx) != 0) { rc = -1000 - t; break; }
        ro_argmax(a, a->q_pin, act);
        ++*a->n_greedy;
      }
      ro_record(a, t, cur, act, 0);                /* rates on the channels of state t (compute_reward_with_channel_selection) */
      ro_spin(&c.workers_done, t + 1);             /* state t + 1 */
      memcpy(a->t_xe_next + (int64_t)t * n * 16, c.st[(t + 1) % ns].xe, (size_t)n * 16 * sizeof(float));
    }
  }
#undef RO_STATE
  const int done = rc == T ? T : (rc <= -1000 ? -1000 - rc : -4 - rc);     /* transitions completed */
  if (team) {
    if (done < T) {
      /* an early exit: let the team run out (it never waits for more than main_pos) and throw its extra steps away -- the
       * environment is committed at state `done`, recomputed below from the stream state of that step */
      atomic_store_explicit(&c.main_pos, T, memory_order_release);
    }
    ro_team_join();
  }
  if (done < T) {
    /* states beyond `done` were computed ahead with draws the environment has not officially taken: redo the committed prefix
     * sequentially from the inputs (cheap next to the failure that brought us here) */
    memcpy(c.mt, a->keys, sizeof(c.mt)); c.mtpos = *a->mtpos;
    memcpy(c.si, a->v2i_shadow, (size_t)n * sizeof(double));
    memcpy(c.sv, a->v2v_shadow, (size_t)n * n * sizeof(double));
    for (int t = 0; t < done; ++t) { ro_stream_step(&c, t); ro_gauss(&c, t, 0, 1); ro_channels(&c, t, 0, 1); ro_observe(&c, t, 0, 1); }
  }
  if (done > 0) {                                  /* commit state `done` to the caller's arrays */
    const ro_state* f = &c.st[done % ns];
    const ro_slot* sl = &c.ring[(done - 1) % RO_RING];
    memcpy(a->keys, c.mt, sizeof(c.mt)); *a->mtpos = c.mtpos;
    memcpy(a->xy, sl->xy, (size_t)n * 2 * sizeof(double));
    memcpy(a->dirs, sl->dirs, (size_t)n);
    memcpy(a->v2i_shadow, c.si, (size_t)n * sizeof(double));
    memcpy(a->v2v_shadow, c.sv, (size_t)n * n * sizeof(double));
    memcpy(a->v2v_abs, f->v2v_abs, (size_t)n * n * sizeof(double));
    memcpy(a->v2i_abs, f->v2i_abs, (size_t)n * sizeof(double));
    memcpy(a->v2v_ff, f->v2v_ff, (size_t)n * n * rb * sizeof(double));
    memcpy(a->v2i_ff, f->v2i_ff, (size_t)n * rb * sizeof(double));
    memcpy(a->interf_db, f->interf_db, (size_t)n * rb * sizeof(double));
    memcpy(a->state, f->state, (size_t)n * W * sizeof(double));
    memcpy(a->adj, f->adj, (size_t)n * n * sizeof(double));
    memcpy(a->xe, f->xe, (size_t)n * 16 * sizeof(float));
    memcpy(a->mask, f->mask, (size_t)n * sizeof(int32_t));
    memcpy(a->col, f->col, (size_t)ne * sizeof(int32_t));
    *a->regular = f->regular;
  }
  a->eps_last = eps;
  pthread_mutex_unlock(&RT.call_mu);
  return rc;
}

/* ---- Memory.sample's draw (BS_brain.py:261): numpy's legacy np.random.choice(n, k, replace=False) --------------------------------
 * = RandomState.permutation(n)[:k] = shuffle(arange(n))[:k]: for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j]), with
 * random_interval's masked rejection on 32-bit outputs (numpy/random/src/distributions/distributions.c; n < 2^32).  The whole
 * memory is permuted for every minibatch -- 1e6 transitions at the reference's capacity -- and numpy spends 15-28 ms on it
 * (memcpy swaps of int64 items behind a Python-level call).  Same generator, same draws, same result here on a caller-provided
 * int32 scratch: key[624] / pos are the process-wide RandomState's (get_state / set_state around the call).  The indices
 * are drawn first (sequential), then the swaps run with the target lines prefetched.                                    */
static inline uint32_t np_interval(uint32_t* mt, int32_t* pos, uint32_t max) {
  uint32_t mask = max, v;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  do { v = mt_next(mt, pos) & mask; } while (v > max);
  return v;
}
/* out[0..k) = np.random.choice(n, k, replace=False) for the generator state (key, pos), which is advanced; scratch: n int32,
 * draws: n uint32.  Returns 0, or -1 for sizes outside 1 <= k <= n < 2^31. */
int v2xsim_np_choice_noreplace(uint32_t* key, int32_t* pos, int64_t n, int64_t k, int32_t* scratch, uint32_t* draws, int64_t* out) {
  if (n < 1 || k < 1 || k > n || n >= ((int64_t)1 << 31)) return -1;
  int32_t p = *pos;
  {
    /* random_interval(i) for i = n-1 .. 1 without a data-dependent branch: every 32-bit output is masked and stored for the
     * current i, which only moves on when the value was accepted (a rejected value is overwritten by the next one);
     * the mask changes where i crosses a power of two */
    int64_t i = n - 1;
    while (i >= 1) {
      uint32_t mask = (uint32_t)i;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
      const int64_t lo = (int64_t)(mask >> 1) + 1;               /* smallest i with this mask */
      while (i >= lo) {
        const uint32_t v = mt_next(key, &p) & mask;
        draws[i] = v;
        i -= v <= (uint32_t)i;
      }
    }
  }
  *pos = p;
  for (int64_t i = 0; i < n; ++i) scratch[i] = (int32_t)i;
  for (int64_t i = n - 1; i >= 1; --i) {
    if (i > 16) __builtin_prefetch(scratch + draws[i - 16], 1, 1);
    const uint32_t j = draws[i];
    const int32_t t = scratch[i];
    scratch[i] = scratch[j];
    scratch[j] = t;
  }
  for (int64_t i = 0; i < k; ++i) out[i] = scratch[i];
  return 0;
}
/* np.random.shuffle(np.arange(n)) consumed without building the array (Model.fit's shuffle of one batch) */
int v2xsim_np_shuffle_skip(uint32_t* key, int32_t* pos, int64_t n) {
  if (n < 0 || n >= ((int64_t)1 << 31)) return -1;
  int32_t p = *pos;
  for (int64_t i = n - 1; i >= 1; --i) (void)np_interval(key, &p, (uint32_t)i);
  *pos = p;
  return 0;
}
