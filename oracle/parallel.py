"""ORACLE -- TEST INFRASTRUCTURE: the compact CPU oracle's fit step on ALL host cores, for bench.py's `cpu_baseline` leg.

The per-slot matrix products of the path (20 products of [4096 x 141] . [141 x 64] per layer at the benchmark's size) are
too small for a BLAS thread pool -- 128 threads on one product are slower than one -- and numpy threads serialise on the
GIL.  What does scale is what the GPU path does across GPUs: graph instances are independent, so W
worker PROCESSES (one BLAS thread each) take contiguous shards of whole graphs, differentiate their share of the global
Huber mean, and the parent sums the W gradients and makes ONE Keras-Adam update.  Parameters and gradient slabs live in
shared memory; the result is oracle/compact.OracleModel.train_step up to the summation order of the gradient.

Follows /root/reference/BS_brain.py:218-223 (one `fit` call on the whole batch) like the rest of the oracle."""
import multiprocessing as mp
from multiprocessing import shared_memory

import numpy as np

from . import compact as oc
from .keras_semantics import KerasAdam
from .spec import GnnSpec


def _views(spec, flat):
    """Parameter structure whose leaves are views into the flat buffer (order of compact.param_arrays)."""
    proto = oc.init_params(spec, np.random.default_rng(0), np.float32)
    pos = 0
    out = {'gnn': [], 'dense': []}
    for st in proto['gnn']:
        d = {}
        for k in ('W1', 'W2', 'W3', 'b'):
            n = st[k].size
            d[k] = flat[pos:pos + n].reshape(st[k].shape)
            pos += n
        out['gnn'].append(d)
    for st in proto['dense']:
        d = {}
        for k in ('W', 'b'):
            n = st[k].size
            d[k] = flat[pos:pos + n].reshape(st[k].shape)
            pos += n
        out['dense'].append(d)
    return out, pos


def _worker(conn, spec_kw, shard, n_graphs_global, rank, world, shm_p, shm_g, n_params):
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    spec = GnnSpec(**spec_kw)
    sp, sg = shared_memory.SharedMemory(name=shm_p), shared_memory.SharedMemory(name=shm_g)
    flat = np.ndarray((n_params,), np.float32, buffer=sp.buf)
    slab = np.ndarray((world, n_params), np.float32, buffer=sg.buf)[rank]
    params, _ = _views(spec, flat)
    xs, es, adj, ys = shard                                              # node rows of this worker's whole graphs
    B = n_graphs_global
    M = oc.csr_to_matrix(*oc.adj_to_csr(adj), dtype=np.float32)
    conn.send("ready")
    while conn.recv() == "step":
        q, cache = oc.forward(spec, params, xs, es, M)
        _, dq = oc.huber_loss_and_grad(spec, q, ys, B)                      # this shard's share of the GLOBAL mean
        g = oc.backward(spec, params, cache, dq)
        pos = 0
        for a in oc.param_arrays(g):
            slab[pos:pos + a.size] = a.ravel()
            pos += a.size
        conn.send("done")
    sp.close()
    sg.close()


class ShardedOracle(object):
    """W worker processes x 1 BLAS thread; step() == one fit step of the whole batch."""

    def __init__(self, spec_kw, x, e, adj, y, workers, seed=1001):
        """x [B, N, Dn], e [B, N, De], adj [B, N, N], y [B * N, C] (bench.synth_batch)"""
        self.spec = GnnSpec(**spec_kw)
        init = oc.init_params(self.spec, np.random.default_rng(seed), np.float32)
        n = sum(a.size for a in oc.param_arrays(init))
        self.n_params, self.workers = n, workers
        self.shm_p = shared_memory.SharedMemory(create=True, size=4 * n)
        self.shm_g = shared_memory.SharedMemory(create=True, size=4 * n * workers)
        self.flat = np.ndarray((n,), np.float32, buffer=self.shm_p.buf)
        self.slabs = np.ndarray((workers, n), np.float32, buffer=self.shm_g.buf)
        self.flat[:] = np.concatenate([a.ravel() for a in oc.param_arrays(init)])
        self.opt = KerasAdam()
        ctx = mp.get_context("spawn")              # not fork: the parent may hold an initialised HIP runtime
        self.conns, self.procs = [], []
        # one BLAS thread per worker, also when threadpoolctl is missing: the spawned interpreters read these when they
        # import numpy (`workers` x the full BLAS pool would understate the all-cores leg)
        import os
        saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        for k in saved:
            os.environ[k] = "1"
        B, N = x.shape[0], x.shape[1]
        for r in range(workers):
            g0, g1 = r * B // workers, (r + 1) * B // workers
            shard = (x[g0:g1].reshape((g1 - g0) * N, -1), e[g0:g1].reshape((g1 - g0) * N, -1), adj[g0:g1],
                     y.reshape(B, N, -1)[g0:g1].reshape((g1 - g0) * N, -1))
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(b, spec_kw, shard, B, r, workers, self.shm_p.name, self.shm_g.name, n),
                            daemon=True)
            p.start()
            self.conns.append(a)
            self.procs.append(p)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        try:
            self._expect("ready", 120.0)
        except Exception:
            self.close()
            raise

    def _expect(self, word, timeout_s):
        """every worker answers `word` within timeout_s; a worker that died or hangs raises instead of blocking the caller"""
        import time
        deadline = time.monotonic() + timeout_s
        for c, p in zip(self.conns, self.procs):
            while not c.poll(0.2):
                if not p.is_alive():
                    raise RuntimeError("oracle worker %d exited with code %s" % (p.pid, p.exitcode))
                if time.monotonic() > deadline:
                    raise RuntimeError("oracle worker %d did not answer within %.0f s" % (p.pid, timeout_s))
            got = c.recv()
            if got != word:
                raise RuntimeError("oracle worker %d answered %r, expected %r" % (p.pid, got, word))

    def step(self):
        for c in self.conns:
            c.send("step")
        self._expect("done", 600.0)
        g = self.slabs.sum(axis=0)
        self.opt.step([self.flat], [g])

    def close(self):
        for c in self.conns:
            try:
                c.send("stop")
            except Exception:
                pass
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        for s in (self.shm_p, self.shm_g):
            try:
                s.close()
                s.unlink()
            except Exception:
                pass
