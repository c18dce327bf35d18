/* libv2xsim.so -- host-side helper of the CALLERS of the hot path (the batched simulator behind the DQN loop),
 * not part of the drop-in boundary (that is include/v2xgnn.h).  Plain C, no GPU: the array arithmetic of E independent V2X
 * simulators, one environment per task of the library's own thread pool (threads sleep between jobs).  Bound with ctypes in
 * globecom2020-resourceallocationgnn_amd/rl/native_sim.py; the numpy expressions of rl/batched_env.py are the definition and
 * the fallback, tests/test_rl_batched_env.py compares the two (integer state and random streams identical, reals to libm
 * rounding).  All arrays are C-contiguous; E environments, n vehicles = links, rb resource blocks.
 * MT19937 states are numpy RandomState layout: keys[E][624] uint32 + pos[E] int32, advanced in place.
 * One caller thread at a time.                                                                                           */
#ifndef V2XSIM_H
#define V2XSIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int v2xsim_abi(void);                       /* 3 */
void v2xsim_set_threads(int n);             /* threads a loop may use, the caller included */
int v2xsim_max_threads(void);

/* renew_channel + renew_channels_fastfading (Environment.py:378-406, path loss :94-146) from the step's uniforms u[E][n_u]
 * (random.gauss order: cos value, then the sin value of a pair); shadowing states are updated in place */
void v2xsim_channels(int E, int n, int rb, const double* u, int n_u, const double* vel, const double* pos, double* v2i_shadow,
                     double* v2v_shadow, double* v2v_abs, double* v2i_abs, double* v2v_ff, double* v2i_ff, double* scratch);
/* compute_reward_with_channel_selection (Environment.py:408-458), every link active, one receiver per link */
void v2xsim_reward(int E, int n, int rb, const int64_t* ch, const int64_t* dest, const double* v2v_ff, const double* v2i_ff,
                   const double* v2i_abs, double p_v2v, double p_v2i, double veh_gain, double bs_gain, double bs_nf, double veh_nf,
                   double sig2, double* v2v_rate, double* v2i_rate, double* interference, double* v2i_interf, double* v2v_interf);
/* Compute_Interference (Environment.py:460-493), the observable part, in dB: out[E][n][rb] */
void v2xsim_interference(int E, int n, int rb, const int64_t* dest, const double* v2v_ff, double p_v2i, double veh_gain,
                         double veh_nf, double sig2, double* out);
/* Agent.observe for all environments (BS_brain.py:389-407, :441-445, :458-467): state[E][n][3C+1], adj[E][n][n] */
void v2xsim_observe(int E, int n, int C, const int64_t* dest, const double* v2v_ff, const double* v2i_ff, double power,
                    double* state, double* adj);
/* ... and the same observation in the engine's packed layout (include/v2xgnn.h): xe[E][n][16] float32, source masks
 * mask[E][n], CSR sources col[E][n (n-2)] (zeros for a graph with a link that is its own receiver), regular[E] */
void v2xsim_observe_packed(int E, int n, int C, const int64_t* dest, const double* v2v_ff, const double* v2i_ff, double power,
                           double* state, double* adj, float* xe, int32_t* mask, int32_t* col, uint8_t* regular);
/* the next n_u random.random() doubles of every stream */
void v2xsim_mt_uniforms(int E, uint32_t* keys, int32_t* pos, double* out, int n_u);
/* add_new_vehicles_by_number (Environment.py:217-234): the integer draws of an episode reset in the stdlib generator's order */
void v2xsim_reset_vehicles(int E, int n, uint32_t* keys, int32_t* pos, int n_lanes, const double* down, const double* up,
                           const double* left, const double* right, int width, int height, double* xy, int8_t* dirs, double* vel);
/* renew_neighbor's draw (Environment.py:375): random.sample(candidates, 1)[0] per link, populations of 1..21 */
void v2xsim_sample_dest(int E, int n, int m, uint32_t* keys, int32_t* pos, const int64_t* cand, int64_t* dest);
/* renew_positions (Environment.py:236-345) in place; dirs: 0 up, 1 down, 2 left, 3 right */
void v2xsim_positions(int E, int n, uint32_t* keys, int32_t* pos, double* xy, int8_t* dirs, const double* vel, double timestep,
                      int n_lanes, const double* up, const double* down, const double* left, const double* right, double width,
                      double height);

/* One whole simulator step (what Agent.act runs after the rates, BS_brain.py:366-376: positions, channels, interference) plus the
 * next observation, as a map state-in -> state-out that never writes its inputs.  v2xsim_advance runs it now;
 * v2xsim_advance_start on the pool alone while the caller does something else (> 0: the job's ticket; -1: another job is still
 * running -- one that is done and was never waited for is replaced; -2: no thread), v2xsim_advance_wait(ticket) returns when
 * that job is done (ticket 0: whatever is in flight).  One job in flight per process.  rl/batched_env.py (`lookahead`). */
typedef struct {
  int32_t E, n, rb, n_lanes;
  double timestep, width, height;
  const double *up, *down, *left, *right;
  const double* vel; const int64_t* dest;
  double p_v2v, p_v2i, veh_gain, veh_nf, sig2;
  const uint32_t* keys_in; const int32_t* mtpos_in; const double* xy_in; const int8_t* dirs_in;
  const double* v2i_shadow_in; const double* v2v_shadow_in;
  uint32_t* keys; int32_t* mtpos; double* xy; int8_t* dirs; double* v2i_shadow; double* v2v_shadow;
  double *v2v_abs, *v2i_abs, *v2v_ff, *v2i_ff, *interf_db, *state, *adj;
  float* xe; int32_t* mask; int32_t* col; uint8_t* regular;
  double* scratch;                           /* [E][2 n_u], n_u = n + n^2 + 2 n rb + 2 n^2 rb (even) */
} v2xsim_advance_args;
void v2xsim_advance(const v2xsim_advance_args* a);
int v2xsim_advance_start(const v2xsim_advance_args* a);
int v2xsim_advance_wait(int ticket);

/* The reference's own rollout shape as ONE call (Agent.generate_d2d_transition, BS_brain.py:409-553, called with 50 transitions
 * before every replay, :818-832): ONE simulator, T sequential transitions -- epsilon draw / random actions on numpy's process-wide
 * MT19937 (np_key / np_pos: get_state / set_state around the call; random_sample and randint(0, n_actions) draw for draw), or the
 * predict through `predict(predict_ctx)` (the caller's closure scores the ONE graph whose packed observation this call has put
 * into xe_pin [n][16] / col_pin [n (n-2)] and leaves Q in q_pin [n][n_actions]; 0 = ok) with np.argmax's first maximiser;
 * v2xsim_reward's rates on the current channels; one v2xsim_advance step -- cut over a team of threads (v2xsim_set_threads)
 * that work ahead of the caller while it waits for the predict.  The environment (E = 1 arrays of v2xsim_advance_args, here
 * updated IN PLACE to the state after the last transition) and both random streams end exactly where T single steps leave
 * them; the t_* arrays receive the T transitions.  Returns T; -1 bad sizes; -2 no memory; -3 another rollout is running;
 * -4 - t: transition t wanted a predict on a graph with a link that is its own receiver (regular == 0) or predict is null;
 * -1000 - t: the predict of transition t failed.  After an early return the environment stands at the state after the transitions
 * that were completed (t of them), numpy's stream after the failed transition's epsilon draw.
 * batch_predict = 1: nothing in a simulator step depends on the actions and the network does not change inside a rollout, so the
 * T observations are all known before any action is: the team computes the T states while the caller takes the policy draws (a
 * greedy transition draws nothing but its epsilon), then `predict` is called ONCE for all T graphs (xe_pin [T][n][16], col_pin
 * [T][n (n-2)], q_pin [T][n][n_actions]; a graph's Q-values do not depend on the batch around it), then argmax / rates / records
 * transition by transition.  Same transitions, same streams; a failing batch predict counts as a failure of the first greedy
 * transition.                                                                                                              */
typedef int (*v2xsim_predict_fn)(void* ctx);
typedef struct {
  int32_t n, rb, n_lanes, T, n_actions, batch_predict;
  double timestep, width, height;
  const double *up, *down, *left, *right;
  const double* vel; const int64_t* dest;
  double p_v2v, p_v2i, veh_gain, veh_nf, sig2, bs_gain, bs_nf;
  /* the environment, in and out */
  uint32_t* keys; int32_t* mtpos; double* xy; int8_t* dirs; double* v2i_shadow; double* v2v_shadow;
  double *v2v_abs, *v2i_abs, *v2v_ff, *v2i_ff, *interf_db, *state, *adj;
  float* xe; int32_t* mask; int32_t* col; uint8_t* regular;
  double *interference, *v2i_interf, *v2v_interf;      /* [rb], [rb], [n]: v2xsim_reward's side outputs of the LAST transition */
  /* the policy */
  uint32_t* np_key; int32_t* np_pos;
  double eps_max, eps_min, eps_per_step, eps_steps; int64_t step_no0;
  v2xsim_predict_fn predict; void* predict_ctx; float* xe_pin; int32_t* col_pin; const float* q_pin;
  /* out: the T transitions, and what the agent keeps */
  float* t_xe; float* t_xe_next; int32_t* t_col; int32_t* t_mask; uint8_t* t_regular; int64_t* t_action;
  double* t_v2v_rate; double* t_v2i_rate;               /* [T][n], [T][min(rb, n)] */
  int32_t* n_greedy; double eps_last;
} v2xsim_rollout_args;
int v2xsim_rollout(v2xsim_rollout_args* a);

/* The epsilon-greedy draws of one iteration over E simulators (BS_brain.py:308-352 per simulator, in order) on numpy's process-wide
 * MT19937: epsilon of step step_no0 + e, one random_sample(); below epsilon n randint(0, n_actions) draws -> actions[e][0..n) and
 * greedy[e] = 0, else greedy[e] = 1.  Returns the number of greedy simulators (-1: bad argument); *eps_last = the last epsilon. */
int v2xsim_np_policy_draws(uint32_t* np_key, int32_t* np_pos, int32_t E, int32_t n, int32_t n_actions, double eps_max, double eps_min,
                           double eps_per_step, double eps_steps, int64_t step_no0, int64_t* actions, uint8_t* greedy, double* eps_last);

/* Memory.sample's draw (BS_brain.py:261): numpy's legacy np.random.choice(n, k, replace=False) = permutation(n)[:k] on the
 * process-wide RandomState's MT19937 state (key[624], pos: get_state / set_state around the call), draw for draw; scratch [n]
 * int32, draws [n] uint32; 0 or -1 (sizes).  v2xsim_np_shuffle_skip: the draws of np.random.shuffle(np.arange(n)) only. */
int v2xsim_np_choice_noreplace(uint32_t* key, int32_t* pos, int64_t n, int64_t k, int32_t* scratch, uint32_t* draws, int64_t* out);
int v2xsim_np_shuffle_skip(uint32_t* key, int32_t* pos, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
