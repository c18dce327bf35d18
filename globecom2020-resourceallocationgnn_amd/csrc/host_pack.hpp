// Host side of the drop-in boundary: the reference's dict payload -> the packed batch of include/v2xgnn.h in ONE pass of
// compiled code (v2x_pack_feed).
//
// What the reference hands to Model.predict / Model.fit (BS_brain.py:495-504, :642-651, :704-716) is, per call, 3 N
// arrays [B, width] ('D{k}_Node_Input', 'D{k}_Edge_Input', 'D{k}_Neighbor_Input', float64) and the dense
// 'Adjacency_Matrix' [B, N F, N F] = kron(Adj, I_F) (:492-493, :603) -- 32 KB per graph at the reference's own
// configuration (4 links x 16 features), 16.8 MB per replay minibatch of 512.  packing.py's numpy version of this
// (feed_to_arrays + kron_to_adj + adj_to_csr + pack_xe; it stays the definition and what the tests compare against) walks
// that array four times (einsum, two count_nonzero, a strided copy) and cost 5.2 ms per call -- more than a whole fit
// step of the CPU restatement.  Here: one read of the adjacency (graphs split over a few threads), the Kronecker
// structure checked while it streams by, CSR by destination and the [x | e | pad] rows written directly.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

namespace v2x_host {

enum PackErr { PACK_OK = 0, PACK_NOT_KRON = 1, PACK_NOT_01 = 2 };

template <typename T> struct Bits;
template <> struct Bits<double> { typedef uint64_t U; static constexpr U ONE = 0x3FF0000000000000ull; };
template <> struct Bits<float> { typedef uint32_t U; static constexpr U ONE = 0x3F800000u; };

// Number of entries of v[0..n) that are not +-0 (NaN counts as non-zero: it is then rejected as "not 0 or 1" or as
// an off-diagonal entry).  Written on the bit patterns so that the loop vectorises without floating-point compares.
template <typename T>
inline long count_nonzero_bits(const T* v, long n) {
  typedef typename Bits<T>::U U;
  const U* u = reinterpret_cast<const U*>(v);
  long c = 0;
  for (long i = 0; i < n; ++i) c += (U)(u[i] << 1) != 0;
  return c;
}

// One graph: A[NF][NF] must be kron(Adj, I_F) with Adj in {0, 1}.  adj_out[q * N + p] = Adj[p][q] (destination-major).
// The full check ORs every entry of a row that is NOT on a block diagonal (through an F-entry mask that is the same for
// every block of the row: plain and / or on the bit patterns, which vectorises with baseline SSE2) and compares the
// block-diagonal entries with the strided sample A[p F][q F] the engine uses.
template <typename T>
inline int scan_graph(const T* A, int N, int F, bool check, const typename Bits<T>::U* offdiag_mask, unsigned char* adj_out) {
  typedef typename Bits<T>::U U;
  const long NF = (long)N * F;
  int err = PACK_OK;
  U bad = 0;
  for (int p = 0; p < N; ++p) {
    const U* r0 = reinterpret_cast<const U*>(A + (long)p * F * NF);
    for (int q = 0; q < N; ++q) {
      const U b = r0[(long)q * F];
      const bool one = b == Bits<T>::ONE;
      if (!one && (U)(b << 1) != 0) err = PACK_NOT_01;
      adj_out[q * N + p] = one ? 1 : 0;
    }
    if (!check) continue;
    for (int i = 0; i < F; ++i) {
      const U* r = r0 + (long)i * NF;
      const U* mi = offdiag_mask + (long)i * F;
      U acc = 0, diff = 0;
      for (int q = 0; q < N; ++q) {
        const U* blk = r + (long)q * F;
        for (int j = 0; j < F; ++j) acc |= blk[j] & mi[j];
        // the block's diagonal is constant: Adj[p][q] for every i -- compared as VALUES (numpy's ==): +0.0 and -0.0 are equal
        // (their patterns differ in the sign bit only), anything non-zero must match bit for bit
        const U xr = blk[i] ^ r0[(long)q * F];
        diff |= (U)(xr << 1) | ((U)(blk[i] << 1) != 0 ? xr : (U)0);
      }
      bad |= (U)(acc << 1) | diff;                    // +-0 off the diagonals; NaN or anything else is a violation
    }
  }
  if (bad != 0 && !err) err = PACK_NOT_KRON;
  return err;
}

template <typename T>
inline void copy_row(float* dst, const T* src, int n) {
  for (int i = 0; i < n; ++i) dst[i] = (float)src[i];
}

struct FeedView {
  int B, N, F, Dn, De;
  const void* const* node; const void* const* edge; const void* const* nbr;
  const unsigned char* is_f64;          // [3 N + 1]: node[0..N), edge[0..N), nbr[0..N), adjacency
  const void* adjacency;
};

// A few persistent worker threads for the adjacency scan.  Starting a std::thread costs ~30 us on the GPU hosts (measured:
// 16 fresh threads per call = 450 us, more than the scan they were meant to speed up); parked workers wake in a few us.
// Created on first use, never joined (the process exits under them); a fork()ed child starts with an empty pool.
class WorkerPool {
 public:
  static WorkerPool& get() {
    static WorkerPool* pool = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
      pool = new WorkerPool();
      pthread_atfork(nullptr, nullptr, [] { pool->after_fork(); });
    });
    return *pool;
  }
  // runs fn(0..n_parts) on the caller + up to n_parts - 1 workers; returns when all parts are done
  void run(int n_parts, const std::function<void(int)>& fn) {
    if (n_parts <= 1) { fn(0); return; }
    std::unique_lock<std::mutex> call(call_mu_);                 // one parallel region at a time
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)threads_.size() < n_parts - 1) {
        const int id = (int)threads_.size();
        threads_.emplace_back([this, id] { loop(id); });
        threads_.back().detach();
      }
      fn_ = &fn; n_parts_ = n_parts; pending_ = n_parts - 1; ++gen_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(int id) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (id + 1 < n_parts_) fn = fn_;
      }
      if (!fn) continue;
      (*fn)(id + 1);
      std::unique_lock<std::mutex> lk(mu_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  void after_fork() {                                             // child: the workers do not exist here
    new (&mu_) std::mutex(); new (&call_mu_) std::mutex();
    new (&cv_) std::condition_variable(); new (&done_) std::condition_variable();
    new (&threads_) std::vector<std::thread>();
    fn_ = nullptr; pending_ = 0; n_parts_ = 0;
  }
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_parts_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
};

// threads for a scan of `bytes` bytes over n_graphs graphs: >= 1 MB each, at most 8 (V2X_PACK_THREADS overrides)
inline int n_pack_threads(int n_graphs, size_t bytes) {
  static const int env = getenv("V2X_PACK_THREADS") ? atoi(getenv("V2X_PACK_THREADS")) : 0;
  int t;
  if (env > 0) t = env;
  else {
    t = (int)std::thread::hardware_concurrency();
    if (t > 8) t = 8;
    const int by_size = (int)(bytes >> 20);
    if (t > by_size) t = by_size;
  }
  if (t > n_graphs) t = n_graphs;
  if (t > 64) t = 64;
  return t < 1 ? 1 : t;
}

// -> 0, or the PackErr of the first offending graph (*bad_graph).  xe [B N][16], row_ptr [B N + 1], col_idx [<= B N N],
// nbr_out [B N][F] (written only when some Neighbor_Input entry is non-zero: info[2]), info = {n_edges, max_edges, nbr_nonzero}
inline int pack_feed(const FeedView& f, bool check_kron, float* xe, int32_t* row_ptr, int32_t* col_idx, float* nbr_out,
                     int32_t* info, int* bad_graph) {
  const int B = f.B, N = f.N, F = f.F, XEW = 16;
  const long NF = (long)N * F;
  const bool a64 = f.is_f64[3 * N] != 0;
  const size_t a_bytes = (size_t)B * NF * NF * (a64 ? 8 : 4);
  std::vector<unsigned char> adjc((size_t)B * N * N);
  std::vector<uint64_t> mask64;
  std::vector<uint32_t> mask32;
  if (check_kron) {                                   // mask[i][j] = all ones except j == i
    if (a64) { mask64.assign((size_t)F * F, ~0ull); for (int i = 0; i < F; ++i) mask64[(size_t)i * F + i] = 0; }
    else { mask32.assign((size_t)F * F, ~0u); for (int i = 0; i < F; ++i) mask32[(size_t)i * F + i] = 0; }
  }
  // without the full check only the strided sample A[b][p F][q F] is read: nothing worth splitting
  const int T = check_kron ? n_pack_threads(B, a_bytes) : 1;
  std::vector<int> t_err(T, 0), t_bad(T, B), t_any(T, 0);
  auto work = [&](int t, int g0, int g1) {
    int err = 0, bad = B, any = 0;
    for (int b = g0; b < g1; ++b) {
      int e;
      if (a64) e = scan_graph(static_cast<const double*>(f.adjacency) + (size_t)b * NF * NF, N, F, check_kron, mask64.data(), &adjc[(size_t)b * N * N]);
      else e = scan_graph(static_cast<const float*>(f.adjacency) + (size_t)b * NF * NF, N, F, check_kron, mask32.data(), &adjc[(size_t)b * N * N]);
      if (e && !err) { err = e; bad = b; }
      for (int k = 0; k < N; ++k) {
        float* dst = xe + ((size_t)b * N + k) * XEW;
        if (f.is_f64[k]) copy_row(dst, static_cast<const double*>(f.node[k]) + (size_t)b * f.Dn, f.Dn);
        else copy_row(dst, static_cast<const float*>(f.node[k]) + (size_t)b * f.Dn, f.Dn);
        if (f.is_f64[N + k]) copy_row(dst + f.Dn, static_cast<const double*>(f.edge[k]) + (size_t)b * f.De, f.De);
        else copy_row(dst + f.Dn, static_cast<const float*>(f.edge[k]) + (size_t)b * f.De, f.De);
        for (int i = f.Dn + f.De; i < XEW; ++i) dst[i] = 0.f;
        if (f.nbr && !any) {
          any = f.is_f64[2 * N + k] ? count_nonzero_bits(static_cast<const double*>(f.nbr[k]) + (size_t)b * F, F) != 0
                                    : count_nonzero_bits(static_cast<const float*>(f.nbr[k]) + (size_t)b * F, F) != 0;
        }
      }
    }
    t_err[t] = err; t_bad[t] = bad; t_any[t] = any;
  };
  WorkerPool::get().run(T, [&](int t) { work(t, (int)((long)B * t / T), (int)((long)B * (t + 1) / T)); });
  int nbr_any = 0;
  for (int t = 0; t < T; ++t) {                       // threads own ascending graph ranges: the first error is the lowest graph
    nbr_any |= t_any[t];
    if (t_err[t]) { *bad_graph = t_bad[t]; return t_err[t]; }
  }
  // CSR by destination: sources of row (b, q) ascending
  int32_t e = 0, max_e = 0;
  row_ptr[0] = 0;
  for (int b = 0; b < B; ++b) {
    const int32_t e0 = e;
    for (int q = 0; q < N; ++q) {
      const unsigned char* a = &adjc[((size_t)b * N + q) * N];
      for (int p = 0; p < N; ++p)
        if (a[p]) col_idx[e++] = p;
      row_ptr[(size_t)b * N + q + 1] = e;
    }
    if (e - e0 > max_e) max_e = e - e0;
  }
  info[0] = e; info[1] = max_e; info[2] = nbr_any;
  if (nbr_any && nbr_out) {
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < N; ++k) {
        float* dst = nbr_out + ((size_t)b * N + k) * F;
        if (f.is_f64[2 * N + k]) copy_row(dst, static_cast<const double*>(f.nbr[k]) + (size_t)b * F, F);
        else copy_row(dst, static_cast<const float*>(f.nbr[k]) + (size_t)b * F, F);
      }
  }
  return PACK_OK;
}

}  // namespace v2x_host
