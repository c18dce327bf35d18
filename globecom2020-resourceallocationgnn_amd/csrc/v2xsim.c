/* libv2xsim.so -- the array arithmetic of one batched simulator step, one OpenMP thread per group of environments.
 *
 * Counterpart of rl/batched_env.py's numpy expressions (which restate /root/reference/Environment.py:94-146 path loss,
 * :378-393 shadowing, :395-406 Rayleigh fast fading, :408-493 rates / interference and BS_brain.py:389-467 observation)
 * for E independent environments of n vehicles / links.  The random streams stay in Python (MT19937 per environment with
 * the stdlib's draw order, rl/mtstream.py): this library receives the UNIFORMS of a step and turns them into Gaussians
 * in random.gauss order (cos value, then the sin value of the same pair).  Same formulas, same operation order as the
 * numpy code; results agree to the last bits of libm (the tests compare at 1e-12).
 * Plain C, no Python API: bound with ctypes (rl/native_sim.py).  gcc -O2 -fopenmp -shared -fPIC v2xsim.c -lm        */
#include <math.h>
#include <stdint.h>

#define TWOPI 6.283185307179586476925286766559

static const double V2V_H = 1.5, FC = 2.0, V2V_DECORR = 10.0, V2V_SHADOW_STD = 3.0;
static const double V2I_H_BS = 25.0, V2I_H_MS = 1.5, V2I_DECORR = 50.0, V2I_SHADOW_STD = 8.0;
static const double BS_X = 750.0 / 2, BS_Y = 1299.0 / 2;

int v2xsim_abi(void) { return 1; }

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
void v2xsim_channels(int E, int n, int rb, const double* u, int n_u, const double* vel, const double* pos,
                     double* v2i_shadow, double* v2v_shadow, double* v2v_abs, double* v2i_abs, double* v2v_ff,
                     double* v2i_ff, double* scratch /* [E][n_u] */) {
  const int n_sh = n + n * n, a = n * rb, b = n * n * rb;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    const double* ue = u + (int64_t)e * n_u;
    double* g = scratch + (int64_t)e * n_u;
    for (int k = 0; k < n_u; k += 2) {
      const double x2pi = ue[k] * TWOPI;
      const double g2rad = sqrt(-2.0 * log(1.0 - ue[k + 1]));
      g[k] = cos(x2pi) * g2rad;
      g[k + 1] = sin(x2pi) * g2rad;
    }
    const double* ve = vel + (int64_t)e * n;
    const double* pe = pos + (int64_t)e * n * 2;
    double* si = v2i_shadow + (int64_t)e * n;
    double* sv = v2v_shadow + (int64_t)e * n * n;
    double* av = v2v_abs + (int64_t)e * n * n;
    double* ai = v2i_abs + (int64_t)e * n;
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
    double* fi = v2i_ff + (int64_t)e * a;
    for (int k = 0; k < a; ++k) {                  /* 20 log10 |(re + j im) / sqrt 2| */
      const double re = rs2 * f[k], im = rs2 * f[a + k];
      fi[k] = ai[k / rb] - 20 * log10(hypot(re, im));
    }
    double* fv = v2v_ff + (int64_t)e * b;
    for (int k = 0; k < b; ++k) {
      const double re = rs2 * f[2 * a + k], im = rs2 * f[2 * a + b + k];
      fv[k] = av[k / rb] - 20 * log10(hypot(re, im));
    }
  }
}

/* compute_reward_with_channel_selection (Environment.py:408-458; every link active, one receiver per link).
 * ch[E][n] chosen resource block, dest[E][n] receiver of link k; out: v2v_rate[E][n], v2i_rate[E][m], m = min(rb, n),
 * interference[E][rb] (without noise), v2i_interf[E][rb] and v2v_interf[E][n] (with noise).                          */
void v2xsim_reward(int E, int n, int rb, const int64_t* ch, const int64_t* dest, const double* v2v_ff, const double* v2i_ff,
                   const double* v2i_abs, double p_v2v, double p_v2i, double veh_gain, double bs_gain, double bs_nf,
                   double veh_nf, double sig2, double* v2v_rate, double* v2i_rate, double* interference,
                   double* v2i_interf, double* v2v_interf) {
  const int m = rb < n ? rb : n;
  const double gain = 2 * veh_gain - veh_nf;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    const int64_t* c = ch + (int64_t)e * n;
    const int64_t* d = dest + (int64_t)e * n;
    const double* vv = v2v_ff + (int64_t)e * n * n * rb;
    const double* vi = v2i_ff + (int64_t)e * n * rb;
    double* itf = interference + (int64_t)e * rb;
    for (int r = 0; r < rb; ++r) itf[r] = 0.0;
    for (int k = 0; k < n; ++k)                    /* (at_bs * onehot).sum(axis=1): ascending k per block */
      itf[c[k]] += pow(10.0, (p_v2v - vi[k * rb + c[k]] + veh_gain + bs_gain - bs_nf) / 10);
    for (int r = 0; r < rb; ++r) v2i_interf[(int64_t)e * rb + r] = itf[r] + sig2;
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
      v2v_interf[(int64_t)e * n + k] = tot;
      v2v_rate[(int64_t)e * n + k] = log2(1 + signal / tot);
    }
    for (int k = 0; k < m; ++k) {
      const double s = p_v2i - v2i_abs[(int64_t)e * n + k] + veh_gain + bs_gain - bs_nf;
      v2i_rate[(int64_t)e * m + k] = log2(1 + pow(10.0, s / 10) / v2i_interf[(int64_t)e * rb + k]);
    }
  }
}

/* Compute_Interference (Environment.py:460-493, observable part): out[E][n][rb] in dB */
void v2xsim_interference(int E, int n, int rb, const int64_t* dest, const double* v2v_ff, double p_v2i, double veh_gain,
                         double veh_nf, double sig2, double* out) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    const double* vv = v2v_ff + (int64_t)e * n * n * rb;
    for (int k = 0; k < n; ++k) {
      const int64_t rx = dest[(int64_t)e * n + k];
      for (int r = 0; r < rb; ++r) {
        double v = sig2;
        /* numpy indexes vehicle number r as the block's V2I transmitter; r < n is the caller's precondition (rb <= n) */
        v += pow(10.0, (p_v2i - vv[((int64_t)r * n + rx) * rb + r] + 2 * veh_gain - veh_nf) / 10);
        out[((int64_t)e * n + k) * rb + r] = 10 * log10(v);
      }
    }
  }
}

/* Agent.observe for all environments (BS_brain.py:389-407, :441-445, :458-467): state[E][n][3C+1], adj[E][n][n] */
void v2xsim_observe(int E, int n, int C, const int64_t* dest, const double* v2v_ff, const double* v2i_ff, double power,
                    double* state, double* adj) {
  const double A = 80, Bc = 60;
  const int W = 3 * C + 1;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    const double* vv = v2v_ff + (int64_t)e * n * n * C;
    const double* vi = v2i_ff + (int64_t)e * n * C;
    const int64_t* d = dest + (int64_t)e * n;
    double* st = state + (int64_t)e * n * W;
    double* ad = adj + (int64_t)e * n * n;
    for (int p = 0; p < n; ++p)
      for (int q = 0; q < n; ++q) ad[p * n + q] = p == q ? 0.0 : 1.0;
    for (int k = 0; k < n; ++k) {
      const int64_t rx = d[k];
      ad[rx * n + k] = 0.0;
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
    }
  }
}

#ifdef _OPENMP
#include <omp.h>
void v2xsim_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int v2xsim_max_threads(void) { return omp_get_max_threads(); }
#else
void v2xsim_set_threads(int n) { (void)n; }
int v2xsim_max_threads(void) { return 1; }
#endif

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
/* out[e][0..n_u) = the next n_u doubles of stream e */
void v2xsim_mt_uniforms(int E, uint32_t* keys, int32_t* pos, double* out, int n_u) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    uint32_t* mt = keys + (int64_t)e * 624;
    int32_t p = pos[e];
    double* o = out + (int64_t)e * n_u;
    for (int k = 0; k < n_u; ++k) {
      const uint32_t a = mt_next(mt, &p) >> 5, b = mt_next(mt, &p) >> 6;
      o[k] = (a * 67108864.0 + b) / 9007199254740992.0;
    }
    pos[e] = p;
  }
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
void v2xsim_reset_vehicles(int E, int n, uint32_t* keys, int32_t* pos, int n_lanes, const double* down, const double* up,
                           const double* left, const double* right, int width, int height, double* xy, int8_t* dirs,
                           double* vel) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    uint32_t* mt = keys + (int64_t)e * 624;
    int32_t p = pos[e];
    double* x = xy + (int64_t)e * n * 2;
    int8_t* d = dirs + (int64_t)e * n;
    double* v = vel + (int64_t)e * n;
    int k = 0;
    for (int g = 0; g < n / 4; ++g) {
      const uint32_t ind = mt_below(mt, &p, (uint32_t)n_lanes);
      x[2 * k] = down[ind]; x[2 * k + 1] = (double)mt_below(mt, &p, (uint32_t)height + 1); d[k] = 1; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
      x[2 * k] = up[ind];   x[2 * k + 1] = (double)mt_below(mt, &p, (uint32_t)height + 1); d[k] = 0; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
      x[2 * k] = (double)mt_below(mt, &p, (uint32_t)width + 1); x[2 * k + 1] = left[ind];  d[k] = 2; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
      x[2 * k] = (double)mt_below(mt, &p, (uint32_t)width + 1); x[2 * k + 1] = right[ind]; d[k] = 3; v[k] = 10.0 + mt_below(mt, &p, 6); ++k;
    }
    pos[e] = p;
  }
}
/* renew_neighbor's destination draw (Environment.py:375): random.sample(candidates, 1)[0] for every link -- a population
 * of m <= 21 entries takes CPython's pool method, whose first pick is pool[_randbelow(m)].                              */
void v2xsim_sample_dest(int E, int n, int m, uint32_t* keys, int32_t* pos, const int64_t* cand, int64_t* dest) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < E; ++e) {
    uint32_t* mt = keys + (int64_t)e * 624;
    int32_t p = pos[e];
    for (int i = 0; i < n; ++i) dest[(int64_t)e * n + i] = cand[((int64_t)e * n + i) * m + mt_below(mt, &p, (uint32_t)m)];
    pos[e] = p;
  }
}
