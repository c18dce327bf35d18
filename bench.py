#!/usr/bin/env python3
"""Headline benchmark: graph-instances/s of one Model.fit step (forward + Huber + backward + Adam,
plus the RCCL gradient all-reduce when N > 1) of the GNN Q-network on synthetic 20-V2V-link graphs,
feat_dim 64, 2 message-passing layers, batch 4096 per GPU (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel's algorithmic bytes / its average launch duration, measured live
                  with HIP events around every launch of an instrumented (eager) pass of the same steps
  cpu_baseline -- the CPU oracle (a numpy restatement of the reference formulation's math, NOT
                  Keras/TF1) timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
FP32_MFMA_PEAK_TF = 157.3


def synth_batch(rng, B, N, C=4):
    """feature statistics measured from the reference simulator at N=20; topology:
    every link q has one receiver dest[q] != q, edge p->q iff p != q and p != dest[q] (in-degree N-2)."""
    x = np.concatenate([rng.normal(0.84, 0.39, size=(B, N, C)), rng.normal(0.60, 0.21, size=(B, N, C)),
                        np.full((B, N, 1), 10.0)], axis=2).astype(np.float32)
    e = rng.normal(0.88, 0.11, size=(B, N, C)).astype(np.float32)
    dest = rng.integers(0, N - 1, size=(B, N))
    dest = dest + (dest >= np.arange(N)[None, :])
    adj = np.ones((B, N, N), np.float32) - np.eye(N, dtype=np.float32)[None]
    adj[np.repeat(np.arange(B), N), dest.reshape(-1), np.tile(np.arange(N), B)] = 0.0
    y = rng.normal(2.5, 1.0, size=(B * N, C)).astype(np.float32)
    return x, e, adj, y


def synth_ragged(rng, B, lo, hi, C=4):
    """BASELINE config 5: N_g ~ U{lo..hi} links per graph, reference topology (in-degree N_g-2),
    packed with CSR offsets."""
    sizes = rng.integers(lo, hi + 1, size=B)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    R = int(offs[-1])
    cols, deg = [], []
    for n in sizes:
        dest = rng.integers(0, n - 1, size=n)
        dest = dest + (dest >= np.arange(n))
        adj = ~np.eye(n, dtype=bool)
        adj[dest, np.arange(n)] = False
        p, q = np.nonzero(adj.T)                  # rows of adj.T = destinations, ascending sources inside
        cols.append(q.astype(np.int32))
        deg.append(adj.sum(axis=0))
    row_ptr = np.concatenate([[0], np.cumsum(np.concatenate(deg))]).astype(np.int32)
    x = np.concatenate([rng.normal(0.84, 0.39, size=(R, C)), rng.normal(0.60, 0.21, size=(R, C)),
                        np.full((R, 1), 10.0)], axis=1).astype(np.float32)
    e = rng.normal(0.88, 0.11, size=(R, C)).astype(np.float32)
    y = rng.normal(2.5, 1.0, size=(R, C)).astype(np.float32)
    return sizes, offs, row_ptr, np.concatenate(cols), x, e, y


def algorithmic_flops(name, R, F, L, C=4, Dn=9, De=4):
    """fp32 MFMA flops per launch of the dense kernels (SURVEY.md 8(d9) terms x rows); None for kernels that do no
    matrix work (aggregations, Adam)."""
    gnn = 2 * R * (2 * F + Dn + De) * F
    embed = 2 * R * (Dn + De) * F
    dense0 = 2 * R * (2 * F + Dn) * 80
    tail = 2 * R * (80 * 40 + 40 * 20 + 20 * C)
    table = {"k_node_fwd_embed": embed, "k_node_fwd": gnn, "k_wgrad_gnn": L * gnn + embed, "k_wgrad_embed": embed,
             "k_node_dgrad": 2 * R * F * 2 * F,
             "k_dense0_fwd": dense0, "k_wgrad_dense0": dense0, "k_dense0_dgrad": 2 * R * 80 * 2 * F,
             "k_mlp_fwd": (dense0 if F < 128 else 0) + tail,
             "k_mlp_bwd": (2 * R * 80 * 2 * F if F < 128 else 0) + tail,
             "k_wgrad_dense": (dense0 if F < 128 else 0) + tail}
    table["k_mlp_train"] = table["k_mlp_fwd"] + table["k_mlp_bwd"]
    table["k_mlp_train_wg"] = table["k_mlp_train"] + table["k_wgrad_dense"]   # + the four Dense weight gradients, same launch
    table["k_wgrad_all"] = table["k_wgrad_gnn"] + table["k_wgrad_dense"]
    # small batches: Dense-0's weight gradient is a role of the graph layers' launch instead of the MLP launch's (csrc/v2xgnn.hip, dense0_rides)
    table["k_mlp_train_wg123"] = table["k_mlp_train_wg"] - dense0
    table["k_wgrad_gnn_d0"] = table["k_wgrad_gnn"] + dense0
    table["k_gnn_fwd_fused"] = embed + L * gnn                        # embed + L stages (graph-major fused launch)
    table["k_gnn_bwd_fused"] = L * table["k_node_dgrad"]              # L data gradients
    table["k_gnn_fwd_ragged"] = table["k_gnn_fwd_fused"]              # the same layers for variable-size graphs (kernels_ragged.hpp)
    table["k_gnn_bwd_ragged"] = table["k_gnn_bwd_fused"]
    if F >= 128:
        table["k_wgrad_gnn"] = gnn                       # wide path: one launch per stage ...
        table["k_wgrad_wide_all"] = L * gnn + embed + dense0     # ... or all graph layers + Dense-0 as roles of one grid
    return table.get(name)


def algorithmic_bytes(name, B, N, F, E, L=2, C=4, Dn=9, De=4):
    """Unique fp32/int32 bytes a kernel must move per launch (inputs read once + outputs written once;
    weights are cache-resident and not counted) -- the per-graph terms of SURVEY.md 8(d8) x B graphs.
    Fused launches (all GNN weight gradients / all Dense weight gradients) count the sum of their roles."""
    R = B * N
    csr = 4 * E + 4 * (R + 1)
    wg_gnn = 4 * R * (2 * F + Dn + De) + 4 * R * F
    wg_embed = 4 * R * (Dn + De) + 4 * R * F
    table = {
        "k_node_fwd_embed": 4 * R * (Dn + De) + 4 * R * F,
        "k_agg_fwd": 8 * R * F + csr,
        "k_node_fwd": 4 * R * (2 * F + Dn + De) + 4 * R * F,
        "k_mlp_fwd": 4 * R * (Dn + 2 * F) + 4 * R * C,
        "k_mlp_bwd": 4 * R * 2 * C + 4 * R * 2 * F,                 # q, y in; [dh|dagg] out (hidden grads stay on chip in the model)
        "k_agg_bwd": 4 * R * 2 * F + 4 * R * F + 4 * R * F + csr,   # [dh|dagg] + mask in, dpre out
        "k_node_dgrad": 4 * R * F + 4 * R * 2 * F,
        "k_wgrad_gnn": L * wg_gnn + wg_embed,                       # L message-passing stages + embed, one launch
        "k_wgrad_embed": wg_embed,
        "k_wgrad_dense": 4 * R * (Dn + 2 * F + 80) + 4 * R * (80 + 40) + 4 * R * (40 + 20) + 4 * R * (20 + C),
    }
    table["k_wgrad_all"] = table["k_wgrad_gnn"] + table["k_wgrad_dense"]        # every layer's weight gradient, one launch
    # graph-major fused launches (csrc/kernels_fused.hpp): inputs once, every h_s / a_s (dpre_s) once -- the tensors
    # the weight-gradient launch and the decision MLP read; nothing else leaves the CU
    # (the backward gates with the SIGN BITS of h_s the forward leaves behind, one bit per feature, not with the rows)
    table["k_gnn_fwd_fused"] = 4 * R * (Dn + De) + csr + 2 * (L + 1) * 4 * R * F + L * R * F // 8
    table["k_gnn_bwd_fused"] = 4 * R * 2 * F + L * R * F // 8 + csr + (L + 1) * 4 * R * F
    # ragged fused launches (csrc/kernels_ragged.hpp): by-destination / by-source bit masks (4 words per row) instead of the CSR
    # slice, the rows of h_s as ReLU' gates in the backward
    table["k_gnn_fwd_ragged"] = 4 * R * (Dn + De) + 16 * R + 2 * (L + 1) * 4 * R * F
    table["k_gnn_bwd_ragged"] = 4 * R * 2 * F + L * 4 * R * F + 16 * R + (L + 1) * 4 * R * F
    table["k_mlp_train"] = table["k_mlp_fwd"] + table["k_mlp_bwd"] - 4 * R * C  # fwd + Huber + bwd fused: q is not re-read
    table["k_mlp_train_wg"] = table["k_mlp_train"]    # weight gradients from the values on chip: no further node-row bytes
    table["k_mlp_train_wg123"] = table["k_mlp_train"] + 4 * R * 80                       # + the dz1 rows for the Dense-0 role
    table["k_wgrad_gnn_d0"] = table["k_wgrad_gnn"] + 4 * R * (Dn + 2 * F + 80)
    table["k_adj_masks"] = csr + 2 * 4 * R * ((min(max(N, 1), 128) + 31) // 32)   # CSR in, bit masks by source and by destination out
    #                       (ragged batches are modelled as ONE graph of R rows: their masks are 4 words, graphs of <= 128 links)
    if F >= 128:        # wide-feature path (csrc/kernels_wide.hpp): Dense-0 and its gradients are launches of their own,
        #                 Dense 1..3 stay register-chained ("tail" kernels), one weight-gradient launch per graph layer
        table["k_dense0_fwd"] = 4 * R * (Dn + 2 * F) + 4 * R * 80
        table["k_dense0_dgrad"] = 4 * R * 80 + 4 * R * 2 * F
        table["k_wgrad_dense0"] = 4 * R * (Dn + 2 * F + 80)
        table["k_mlp_fwd"] = 4 * R * 80 + 4 * R * (40 + 20 + C)
        table["k_mlp_bwd"] = 4 * R * (80 + 40 + 20 + 2 * C) + 4 * R * (80 + 40 + 20 + C)
        table["k_wgrad_dense"] = 4 * R * (80 + 40) + 4 * R * (40 + 20) + 4 * R * (20 + C)
        table["k_wgrad_gnn"] = wg_gnn
        table["k_wgrad_wide_all"] = L * wg_gnn + wg_embed + table["k_wgrad_dense0"]
    return table.get(name)


def lib_sha256():
    """sha256 of the libv2xgnn.so this process loads (ties measurements kept under profiles/ to the binary they are of)."""
    import hashlib
    from v2xgnn.lib import library_path
    try:
        with open(library_path(), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def src_sha256():
    """sha256 over the kernel and C-ABI sources the library is built from (hipcc output is not bit-reproducible: a rebuild of
    the SAME sources has another sha256 and the same kernels)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "globecom2020-resourceallocationgnn_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pkg, "*.hip")) + glob.glob(os.path.join(pkg, "*.hpp")) + [os.path.join(ROOT, "include", "v2xgnn.h")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def hbm_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled
    per the gfx950 note in MI355X_MICROARCH.md, + WRITE_SIZE).  Counters cannot be read from inside the run, so the
    figure is only reported when it was measured on THIS binary, or on a build of the same kernel sources
    (profiles/hbm_traffic.json records the sha256 of the libv2xgnn.so of its PMC passes and of the sources it was built
    from), and for the default workload; otherwise null."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    default = (args.batch, args.nodes, args.feat, args.layers, args.share_weights, args.ragged) == (4096, 20, 64, 2, False, None)
    if not default or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            rec = json.load(f)
        same_binary = rec.get("lib_sha256") and rec["lib_sha256"] == lib_sha256()
        same_sources = rec.get("src_sha256") and rec["src_sha256"] == src_sha256()
        if not (same_binary or same_sources):
            return None
        return rec.get("bytes_per_launch", {}).get(kernel)
    except Exception:
        return None


def step_bytes_per_graph(N, F, L, E, C=4, Dn=9, De=4):
    """SURVEY.md 8(d8) layer-wise model: step = 3 x forward bytes (256,596 B at cfg-2)."""
    fwd = (4 * N * (Dn + De) + 4 * N * F + (L + 1) * (8 * N * F + 4 * E + 4 * (N + 1))
           + L * (4 * N * (2 * F + Dn + De) + 4 * N * F) + 4 * N * (Dn + 2 * F) + 4 * N * C)
    return 3 * fwd


def usable_cpus():
    """CPUs this process can keep busy: its affinity mask cut to the cgroup's CPU quota (the MI355X boxes show 256 hardware threads and
    grant 16 CPUs' worth of time per 100 ms, `cpu.max` = "1600000 100000": 64 worker processes there are 16 CPUs shared by 64)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed_steps(step, budget_s, min_steps, max_steps, warmup):
    """median seconds per step: `warmup` untimed steps, then >= min_steps (<= max_steps) within ~budget_s"""
    for _ in range(warmup):
        step()
    ts, t_all = [], time.perf_counter()
    while len(ts) < max_steps and (len(ts) < min_steps or time.perf_counter() - t_all < budget_s):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def cpu_baseline(N, F, L, share, B, budget_s=30.0, full=False):
    """The CPU restatements of the path timed on this host, fp32, at 1 BLAS thread and at all
    cores: C0 = the reference's own formulation (per-node weights, dict inputs, dense kron(Adj, I_F) adjacency,
    batched dot: oracle/literal.py) at the reference's configuration N=4, F=16, B=512; C1 = the same formulation at
    20 links x 64 features, B=256 (the dense adjacency is 6.5 MB per graph); C2 = the compact node-row / CSR
    formulation (oracle/compact.py) at the benchmark's own configuration.  `value` is C2 at all cores.  None of them
    is Keras/TF1 (not installable: SURVEY.md 8c); they restate its arithmetic."""
    from oracle import compact as oc, literal as ol
    from oracle.keras_semantics import KerasAdam
    from oracle.spec import GnnSpec as OSpec
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        all_thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits, all_thr = None, len(os.sched_getaffinity(0))
    import contextlib

    def limit(n):
        return threadpool_limits(limits=n) if threadpool_limits is not None else contextlib.nullcontext()

    def literal_leg(n, f, b, min_steps, warm):
        spec = OSpec(n_nodes=n, feat_dim=f, n_mp_layers=2)
        rng = np.random.default_rng(1001)
        x, e, adj, y = synth_batch(rng, b, n)
        feed = {k: v.astype(np.float32) for k, v in ol.feed_from_compact(spec, x, e, adj.astype(np.float32)).items()}
        params = oc.init_params(spec, rng, np.float32)
        yl = [np.ascontiguousarray(y.reshape(b, n, -1)[:, k]) for k in range(n)]
        opt = KerasAdam()
        return lambda: ol.train_step_literal(spec, params, opt, feed, yl), b, min_steps, warm

    def compact_leg(b, min_steps, warm):
        spec = OSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=share)
        rng = np.random.default_rng(1001)
        x, e, adj, y = synth_batch(rng, b, N)
        graph = oc.adj_to_csr(adj)
        om = oc.OracleModel(spec, oc.init_params(spec, rng, np.float32), dtype=np.float32)
        xr, er = x.reshape(b * N, -1), e.reshape(b * N, -1)
        return lambda: om.train_step(xr, er, graph, y), b, min_steps, warm

    def sharded_leg(b, workers):
        """C2 on all cores: `workers` processes x 1 BLAS thread, contiguous shards of whole graphs, gradients summed, one
        Adam update (oracle/parallel.py) -- the formulation the GPU path uses across GPUs."""
        from oracle.parallel import ShardedOracle
        x, e, adj, y = synth_batch(np.random.default_rng(1001), b, N)
        so = ShardedOracle(dict(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=share), x, e, adj, y, workers)
        return so

    plan = [("C0", "reference formulation (dense kron adjacency), N=4 F=16 L=2 B=512", lambda: literal_leg(4, 16, 512, 20, 3)),
            ("C1", "reference formulation (dense kron adjacency), N=20 F=64 L=2 B=256", lambda: literal_leg(20, 64, 256, 3, 1)),
            ("C2", "compact CSR formulation, N=%d F=%d L=%d B=%d, %s weights" % (N, F, L, B, "shared" if share else "per-node"),
             lambda: compact_leg(B, 5, 1))]
    legs, per_leg = {}, budget_s / 7.0
    for name, what, make in plan:
        for label, nthr in (("all_cores", all_thr), ("one_thread", 1)):
            with limit(nthr):
                step, b, min_steps, warm = make()
                if name == "C2" and nthr == 1 and B > 512:         # one thread at the full batch would take minutes
                    step, b, min_steps, warm = compact_leg(512, 3, 1)
                sec, n = _timed_steps(step, per_leg, min_steps, 50, warm)
            legs["%s_%s" % (name, label)] = {"graphs_per_s": round(b / sec, 1), "ms_per_step": round(1e3 * sec, 2), "steps": n,
                                             "batch": b, "threads": int(nthr),
                                             "what": what if b == (B if name == "C2" else b) else what + " (timed on a %d-graph sample)" % b}
    # C2 on all cores for real: the BLAS thread pool cannot use them (the per-slot products are too small: the
    # "all_cores" figure above is SLOWER than one thread on a 128-thread host), worker processes over graph shards can
    host_cores, usable = len(os.sched_getaffinity(0)), usable_cpus()
    # (sized by the hardware threads, not by the quota: measured on a box with 256 threads and a 16-CPU quota, 16 workers 105 k
    #  graphs/s, 32 workers 170 k, 64 workers 159-194 k -- the JSON carries both figures)
    workers = int(os.environ.get("V2X_BENCH_CPU_WORKERS", "0")) or max(1, min(host_cores, B // 64))
    try:
        so = sharded_leg(B, workers)
        try:
            sec, n = _timed_steps(so.step, per_leg, 5, 50, 1)
        finally:
            so.close()
        legs["C2_sharded_processes"] = {"graphs_per_s": round(B / sec, 1), "ms_per_step": round(1e3 * sec, 2), "steps": n, "batch": B,
                                        "threads": int(workers),
                                        "what": "compact CSR formulation, N=%d F=%d L=%d B=%d, %s weights: %d worker processes x 1 BLAS "
                                                "thread over contiguous graph shards, gradients summed, one Adam update"
                                                % (N, F, L, B, "shared" if share else "per-node", workers)}
    except Exception as exc:                       # no /dev/shm, no spawn: keep the other legs
        legs["C2_sharded_processes"] = {"graphs_per_s": 0.0, "error": "%s: %s" % (type(exc).__name__, exc), "threads": int(workers),
                                        "steps": 0, "batch": B}
    # the headline CPU number is the best C2 figure; `cores` = the threads / processes of that run
    main_leg = max(legs["C2_sharded_processes"], legs["C2_all_cores"], legs["C2_one_thread"], key=lambda l: l["graphs_per_s"])
    # `workers` = the threads / processes of that run; `cores` = the CPUs they could keep busy at once (the box's cgroup grants
    # `usable_cpus` CPUs' worth of time however many hardware threads it shows: 64 worker processes there are 16 CPUs shared by 64)
    if not full:
        legs = {k_: {"graphs_per_s": v_["graphs_per_s"], "ms_per_step": v_.get("ms_per_step"), "batch": v_["batch"], "threads": v_["threads"]}
                for k_, v_ in legs.items()}
    return {"value": main_leg["graphs_per_s"], "unit": "graph-instances/s", "usable_cpus": usable, "workers": int(main_leg["threads"]),
            "cores": int(min(main_leg["threads"], usable)), "kind": "port",
            "cpu_model": cpu_model(), "host_cores": host_cores,
            "sample": ("median of %d fit steps of B=%d (N=%d,F=%d,L=%d,%s weights), numpy fp32 CSR oracle (leg C2, %d worker(s)); a CPU "
                       "restatement, not Keras/TF1" % (main_leg["steps"], main_leg["batch"], N, F, L, "shared" if share else "per-node", main_leg["threads"]))
                      + ("; legs: C0 = the reference's formulation (dense kron adjacency) N=4 F=16 B=512, C1 = the same at N=20 F=64 B=256, "
                         "C2 = compact CSR at the benchmark's size" if full else ""),
            "legs": legs}


def rl_episode(links, feat, batch, gamma, train_steps, seed, engine_factory=None, device=0, use_graph=False, envs=0, episodes=1):
    """`episodes` episodes of the DQN loop (RL_Train_main.py:98-118 -> Agent.train, BS_brain.py:750-910) through the package's
    own simulator / agent counterparts: train_steps x (50 rollout transitions + 1 replay of `batch`) each.  The agent is
    built once and runs a two-step warm-up episode first (allocations, hipGraph captures: the reference's Agent also lives
    for the whole run, RL_Train_main.py:92-118); the timed region is Agent.train itself.  -> wall-clock split."""
    import random
    from v2xgnn.rl import Agent, RL_Config
    from v2xgnn.rl.train import start_env, start_env_batched
    random.seed(seed)
    np.random.seed(seed)
    cfg = RL_Config()
    cfg.set_train_value(feat, gamma, batch, 1, 0.1)
    env = start_env_batched(links, envs, seed) if envs > 0 else start_env(links)       # E simulators stepped as arrays
    kw = dict(seed=seed)
    if engine_factory is not None:
        from v2xgnn import BS
        kw = dict(brain=BS(links, 3, 1, feat, env.n_Neighbor, env.n_RB, seed=seed, engine_factory=engine_factory), device_replay=False)
    else:
        kw.update(device=device, use_graph=use_graph)
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, feat, env, cfg, **kw)
    agent.train(1, 2)                                   # warm-up with THIS agent
    split = {"rollout_s": 0.0, "replay_s": 0.0}
    roll, rep = agent.generate_d2d_transition, agent.replay
    rep_dev = getattr(agent, "_replay_on_device", None)

    def timed(fn, key):
        def inner(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            split[key] += time.perf_counter() - t0      # host time of the call: nothing synchronises the GPU here
            return out
        return inner
    agent.generate_d2d_transition, agent.replay = timed(roll, "rollout_s"), timed(rep, "replay_s")
    if rep_dev is not None:
        agent._replay_on_device = timed(rep_dev, "replay_s")
    if engine_factory is None:
        import torch
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = agent.train(episodes, train_steps)[0]
    if engine_factory is None:
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert np.all(np.isfinite(loss))
    n = episodes * train_steps
    return {"wall_s": round(wall, 4), "train_steps": n, "train_steps_per_s": round(n / wall, 3),
            "rollout_s": round(split["rollout_s"], 4), "replay_s": round(split["replay_s"], 4),
            "ms_per_train_step": round(1e3 * wall / n, 3), "rollout_ms_per_step": round(1e3 * split["rollout_s"] / n, 3),
            "replay_ms_per_step": round(1e3 * split["replay_s"] / n, 3),
            "other_ms_per_step": round(1e3 * (wall - split["rollout_s"] - split["replay_s"]) / n, 3),
            "env_steps": int(agent.num_step), "mean_loss": round(float(loss[:, -1].mean()), 6)}


def _print_last(line):
    """The JSON line must be the LAST line on stdout: RCCL writes its version banner through C stdio, which is fully
    buffered on a pipe and would otherwise be flushed after everything Python printed -- drain it first."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(line, flush=True)


def main_rl(args):
    """--workload cfg0: BASELINE configs[0] (default Sim_Config: 4 links, 16 features, batch 256, gamma 0.2, episodes of 20
    train steps), on the engine and -- as cpu_baseline, kind "port" -- on the CPU oracle behind the same agent.
    --workload cfg2loop: configs[2] on one GPU (20 links, 64 features, replay batch 4096, HBM-resident replay)."""
    import torch
    links, feat, batch, gamma = (4, 16, 256, 0.2) if args.workload == "cfg0" else (20, 64, 4096, 0.5)
    steps, episodes = 20, max(1, args.episodes)
    ctx = torch.cuda.stream(torch.cuda.Stream())
    with ctx:
        gpu = rl_episode(links, feat, batch, gamma, steps, 1001, use_graph=not args.no_graph and args.launch != "eager", envs=args.envs, episodes=episodes)
    cpu = None
    if args.workload == "cfg0" and not args.no_cpu_baseline:
        from oracle.engine import OracleEngine
        r = rl_episode(links, feat, batch, gamma, steps, 1001, engine_factory=lambda spec: OracleEngine(spec, dtype=np.float32),
                       envs=args.envs, episodes=1)
        cpu = {"value": r["train_steps_per_s"], "unit": "train-steps/s", "usable_cpus": usable_cpus(), "workers": 1,
               "cores": 1, "kind": "port",
               "cpu_model": cpu_model(), "sample": "one episode (seed 1001, after the same warm-up) with the numpy fp32 oracle as the Q-network", "detail": r}
    _print_last(json.dumps({"metric": "DQN train steps/s (50 simulator transitions + 1 replay of batch %d each), %d V2V links, feat_dim %d"
                                % (batch, links, feat),
                      "value": gpu["train_steps_per_s"], "unit": "train-steps/s", "n_gpus": 1, "steps": steps * episodes, "warmup": 2,
                      "ms_per_step": gpu["ms_per_train_step"], "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded simulator)",
                      "config": {"workload": "BASELINE.json configs[%d]: %d episode(s) x %d train steps x (50 rollouts + 1 replay), "
                                             "%d links, feat_dim=%d, batch %d, gamma %g; one agent, two-step warm-up episode outside "
                                             "the timed region"
                                             % (0 if args.workload == "cfg0" else 2, episodes, steps, links, feat, batch, gamma),
                                 "simulators": ("%d stepped as arrays (rl/batched_env.py)" % args.envs) if args.envs > 0 else "1 (rl/environment.py)",
                                 "split": gpu}, "roofline": None, "cpu_baseline": cpu}))


def dropin_boundary(device, seconds=1.5):
    """The drop-in boundary itself: what one Agent.replay of the reference costs through the dict API --
    BS.predict(states) + BS.predict(states_, target=True) + BS.train_dnn(x, y, B) (BS_brain.py:664-665, :728) on the
    reference's own payload (float64 arrays, dense Adjacency_Matrix = kron(Adj, I_F), zero Neighbor inputs; `x` carries the
    SAME adjacency array object as `states`, `states_` another one with equal content, :603, :623, :716) at the reference's
    configuration N = 4, F = 16, B = 512, and BS.predict_one_step at B = 1 (:336).  The brain is built exactly as
    Agent.__init__ builds it (:300), i.e. eager launches on the default stream.  Wall-clock medians, host packing included."""
    import torch
    from v2xgnn import BS
    N, F, B, C = 4, 16, 512, 4
    rng = np.random.default_rng(1001)
    brain = BS(N, 3, 1, F, 1, C, device=device, seed=7)

    def payload(b):
        x, e, adj, _ = synth_batch(rng, b, N)
        d = {}
        for k in range(N):
            d['D%d_Node_Input' % (k + 1)] = np.ascontiguousarray(x[:, k, :], np.float64)
            d['D%d_Edge_Input' % (k + 1)] = np.ascontiguousarray(e[:, k, :], np.float64)
            d['D%d_Neighbor_Input' % (k + 1)] = np.zeros((b, F))
        d['Adjacency_Matrix'] = np.kron(adj.astype(np.float64), np.eye(F))
        return d, adj
    states, adj = payload(B)
    states_ = dict(payload(B)[0])
    states_['Adjacency_Matrix'] = np.kron(adj.astype(np.float64), np.eye(F))     # test_adjacency_matrix_ = test_adjacency_matrix (:583)
    one = payload(1)[0]

    def replay(parts):
        t0 = time.perf_counter()
        p = brain.predict(states)
        t1 = time.perf_counter()
        p_ = brain.predict(states_, target=True)
        t2 = time.perf_counter()
        y = {}
        for k in range(N):
            t = p[k]
            t[np.arange(B), k % C] = 0.5 + 0.9 * p_[k].max(axis=1)
            y['D%d_Decide_Output' % (k + 1)] = t
        x = dict(states)                                    # a new dict around the same arrays (:704-716)
        t3 = time.perf_counter()
        h = brain.train_dnn(x, y, B)
        t4 = time.perf_counter()
        assert np.isfinite(h.history['loss'][0])
        parts.append((t1 - t0, t2 - t1, t4 - t3))

    def timed(fn, budget):
        for _ in range(3):
            fn()
        ts, t_all = [], time.perf_counter()
        while len(ts) < 2000 and (len(ts) < 10 or time.perf_counter() - t_all < budget):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    parts = []
    timed(lambda: replay(parts), seconds)
    parts = np.array(parts[3:])
    med = np.median(parts, axis=0)
    one_ts = timed(lambda: brain.predict_one_step(one), 0.5)
    # the same triple with the numpy packer of rounds 1-3 (packing.feed_to_arrays + PackedBatch.from_dense), for the record
    from v2xgnn.packing import feed_to_arrays, PackedBatch
    spec = brain.model.spec

    def numpy_pack():
        for d in (states, states_, states):
            xs, es, nbr, a = feed_to_arrays(spec, d, True)
            PackedBatch.from_dense(xs, es, a, nbr)
    np_ts = timed(numpy_pack, 0.5)
    torch.cuda.synchronize()
    return {"what": "Agent.replay through the dict API: BS.predict(states) + BS.predict(states_, target=True) + BS.train_dnn(x, y, B); "
                    "float64 payload, dense kron(Adj, I_F) adjacency (16.8 MB per array), N=4 F=16 B=512; eager, default stream; "
                    "wall clock incl. host packing, Kronecker-structure validation on",
            "replay_triple_ms": round(1e3 * float(np.median(parts.sum(axis=1))), 4),
            "predict_ms": round(1e3 * float(med[0]), 4), "predict_target_ms": round(1e3 * float(med[1]), 4),
            "train_dnn_ms": round(1e3 * float(med[2]), 4), "replays_timed": int(len(parts)),
            "predict_one_step_us": round(1e6 * float(np.median(one_ts)), 2),
            "numpy_packer_alone_ms": round(1e3 * float(np.median(np_ts)), 4),
            "pack_threads": os.environ.get("V2X_PACK_THREADS", "auto (<= 8 pooled workers)")}


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) through
    torch.distributed.run on a free local port and hand its output and exit code through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="graphs in total (strong scaling, the default) / per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default; SURVEY.md 8 d1: the metric's GLOBAL batch is fixed): --batch graphs in total, cut into "
                         "contiguous shards of whole graphs, one per GPU; weak: --batch graphs per GPU (global batch grows with --gpus)")
    ap.add_argument("--shard-of", type=int, default=1, metavar="G",
                    help="single-GPU rehearsal of a G-GPU strong-scaling run: time shard 0 of the global batch cut into G shards "
                         "(Huber mean over the global batch, no all-reduce); value = this shard's graphs / s")
    ap.add_argument("--envs", type=int, default=0, help="cfg0 / cfg2loop: number of simulators stepped as arrays (0 = the single one)")
    ap.add_argument("--episodes", type=int, default=5, help="cfg0 / cfg2loop: timed episodes of 20 train steps")
    ap.add_argument("--no-fast-path", "--no-edge-gather", dest="no_fast_path", action="store_true",
                    help="skip the second timed pass with the complement aggregation (the fast path for complete-minus-few "
                         "graphs) that is printed beside the headline; the headline itself runs the general edge-index gather")
    ap.add_argument("--min-seconds", type=float, default=6.0,
                    help="the timed region is extended to at least this long (more steps than --steps if needed)")
    ap.add_argument("--nodes", type=int, default=20)
    ap.add_argument("--feat", type=int, default=64)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--share-weights", action="store_true", help="one shared weight set instead of the reference's per-node sets")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (= --launch eager)")
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                    help="how a fit step reaches the GPU: one hipGraph replay + the Adam launch, or five eager launches; auto (default) "
                         "times 200 steps of each after the warm-up and keeps the faster (round 6: on ROCm 7.2 a graph replay costs "
                         "3.6-5 us per step more than the eager launches it replaces whenever the host runs ahead of the GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in boundary leg (dict API at the reference's configuration)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short passes of configs[3] / configs[4] at their per-GPU shares (config.other_workloads)")
    ap.add_argument("--other-kernels", action="store_true",
                    help="config.other_workloads in its long form (workload prose, kernel path, per-kernel HIP-event times); the default "
                         "line is kept under 8 KB so that a record that keeps only a tail of stdout keeps all of it")
    ap.add_argument("--no-weak-pass", action="store_true", help="N > 1: skip the second timed pass at --batch graphs PER GPU (config.weak)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--workload", choices=["cfg2", "cfg4", "cfg5", "cfg0", "cfg2loop"], default="cfg2",
                    help="BASELINE.json configs[1] (default, the metric's configuration), [3] (100 links x 256 features x "
                         "3 layers, 8192/8 graphs per GPU) or [4] (8-128 links per graph, shared weights, 16384/8 per GPU)")
    ap.add_argument("--ragged", type=int, nargs=2, metavar=("LO", "HI"), default=None,
                    help="variable-size graphs with LO..HI links (needs --share-weights)")
    return ap


def resolve_workload(args):
    strong = args.scaling == "strong"
    if args.workload == "cfg4":           # BASELINE configs[3]: batch 8192 on 8 GPUs = 1024 per GPU
        args.nodes, args.feat, args.layers, args.batch = 100, 256, 3, (8192 if strong else 1024)
        args.ragged, args.share_weights = None, False
    elif args.workload == "cfg5":         # BASELINE configs[4]: batch 16384 on 8 GPUs = 2048 per GPU
        args.ragged, args.feat, args.layers, args.batch, args.share_weights = [8, 128], 64, 2, (16384 if strong else 2048), True
    if args.shard_of > 1 and (not strong or args.gpus > 1):
        raise SystemExit("--shard-of rehearses ONE shard of a strong-scaling run on one GPU")
    if args.ragged is not None and not args.share_weights:
        raise SystemExit("--ragged needs --share-weights")
    return args


class Ctx(object):
    """process-wide state of a bench run: rank / world / device and the process group"""
    def __init__(self, world, rank, local, dist, force_dp):
        self.world, self.rank, self.local, self.dist, self.force_dp = world, rank, local, dist, force_dp


def make_engine(spec, ctx, args, edge_gather, no_graph=None):
    """edge_gather: the fused graph-layer kernels run the general edge-index gather / segment sum of AggLayer.call
    (V2X_FUSED_COMPL=0 is read when the model is created) instead of the complement rewriting.  The complement engine also
    keeps whole-tile workgroups (V2X_FUSED_SPLIT=0): the split-tile kernels that the library picks for the shares of the
    global batch run the edge form only."""
    from v2xgnn import GnnEngine
    key, val = ("V2X_FUSED_COMPL", "0") if edge_gather else ("V2X_FUSED_SPLIT", "0")
    had = os.environ.get(key)
    if had is None:
        os.environ[key] = val
    try:
        return GnnEngine(spec, device=ctx.local, use_graph=not (args.no_graph if no_graph is None else no_graph))
    finally:
        if had is None:
            del os.environ[key]


def run_workload(args, ctx, light=False):
    """One workload on this process group -> the JSON object of its line.  light: a short pass for config.other_workloads
    (no CPU legs, no second aggregation form)."""
    import torch
    import v2xgnn
    from v2xgnn import GnnSpec, PackedBatch
    from v2xgnn.dp import DataParallelTrainer
    world, rank, local, dist = ctx.world, ctx.rank, ctx.local, ctx.dist
    strong = args.scaling == "strong"
    ragged = args.ragged is not None
    N, F, L, B = args.nodes, args.feat, args.layers, args.batch
    spec = GnnSpec(n_nodes=1 if ragged else N, feat_dim=F, n_mp_layers=L, share_weights=args.share_weights,
                   variable_graphs=ragged)
    # The judged aggregation is the general edge-index gather / segment sum (north_star): the headline engine
    # runs it; the complement rewriting (valid for any adjacency, profitable for complete-minus-few graphs) is the fast path
    # timed beside it.  An explicit V2X_FUSED_COMPL in the environment is respected (one engine, no second pass).
    explicit = os.environ.get("V2X_FUSED_COMPL") is not None
    launch = "eager" if args.no_graph else getattr(args, "launch", "auto")
    eng = make_engine(spec, ctx, args, edge_gather=not explicit, no_graph=launch == "eager")
    wrng = np.random.default_rng(1001)             # identical weights on every rank
    shapes = v2xgnn.keras_list_shapes(spec)
    eng.set_weights([np.zeros(s, np.float32) if len(s) == 1 else
                     wrng.uniform(-np.sqrt(6.0 / sum(s)), np.sqrt(6.0 / sum(s)), size=s).astype(np.float32) for s in shapes])
    if strong and B < world:
        raise SystemExit("--scaling strong: %d graphs cannot be cut into %d shards" % (B, world))

    def make_batch(is_strong, b_graphs):
        """-> (device batch, device targets, sizes, local graphs, local rows, global graphs, Huber denominator, host batch)"""
        # weak: every rank draws its own graphs; strong: every rank draws the SAME global batch and keeps its
        # contiguous shard of whole graphs (equal counts; variable-size graphs: balanced by edges + nodes)
        drng = np.random.default_rng(1001 if is_strong else 1001 + 7919 * rank)
        sizes = None
        if ragged:
            sizes, offs, row_ptr, col_idx, x, e, y = synth_ragged(drng, b_graphs, args.ragged[0], args.ragged[1])
            pb = PackedBatch(b_graphs, 0, v2xgnn.pack_xe(x, e), row_ptr, col_idx, graph_off=offs)
        else:
            if is_strong and b_graphs % world:
                raise SystemExit("--scaling strong: batch %d not divisible by %d GPUs" % (b_graphs, world))
            x, e, adj, y = synth_batch(drng, b_graphs, N)
            pb = PackedBatch.from_dense(x, e, adj)
        n_glob = b_graphs if is_strong else b_graphs * world
        rows_global = pb.n_rows
        n_sh = args.shard_of if args.shard_of > 1 else world
        if is_strong and n_sh > 1:
            pb, (r0, r1) = pb.shard(rank, n_sh, with_rows=True)
            y = y[r0:r1]
            if ragged:
                sizes = np.diff(pb.graph_off)
        dbatch = eng.to_device(pb)
        ydev = torch.from_numpy(np.ascontiguousarray(y)).to(dbatch.device)
        denom = n_glob                                # what the Huber mean divides by: graphs, or node rows when ragged
        if ragged:
            if is_strong:
                denom = rows_global
            else:
                t = torch.tensor([pb.n_rows], dtype=torch.int64, device="cuda")
                if dist is not None:
                    dist.all_reduce(t)
                denom = int(t.item())
        return dbatch, ydev, sizes, pb.n_graphs, pb.n_rows, n_glob, denom, pb

    db, yd, sizes, B_local, n_rows_local, n_global, n_denom, pb = make_batch(strong, B)
    n_shards = args.shard_of if args.shard_of > 1 else world
    use_dp = world > 1 or ctx.force_dp
    trainer = DataParallelTrainer(eng, force=ctx.force_dp) if use_dp else None

    def stepper(engine, tr, dbatch, ydev, denom):
        if tr is not None:
            return lambda: tr.train_step(dbatch, ydev, n_graphs_global=denom, want_loss=False)
        return lambda: engine.train_step(dbatch, ydev, n_global=denom, want_loss=False)

    def timed_region(step, n_steps, min_seconds, warmup):
        """`warmup` untimed steps, then >= n_steps (extended to >= min_seconds) bracketed by barrier + synchronize on both
        sides; elapsed = MAX over ranks.  -> (steps, seconds)"""
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        steps = n_steps
        if min_seconds > 0:
            # a timed region of a few milliseconds is at the mercy of clock ramp-up and host jitter (r01: 20 steps = 8 ms),
            # and the driver's 5-s SMI sampler has to see the GPU busy: probe the step time and extend the run
            # (a probe of 5 steps carried the closing synchronise as a fifth of its time and made the region 12 % short)
            n_probe = 100
            tp = time.perf_counter()
            for _ in range(n_probe):
                step()
            torch.cuda.synchronize()
            probe = (time.perf_counter() - tp) / n_probe
            steps = max(steps, int(np.ceil(1.04 * min_seconds / max(probe, 1e-6))))
            if dist is not None:
                t = torch.tensor([steps], dtype=torch.int64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                steps = int(t.item())
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return steps, el

    stream = torch.cuda.Stream(device=local)
    path = eng.path_info(db)
    E = pb.n_edges

    # ---- launch form (--launch auto): the same steps as ONE hipGraph replay + the Adam launch and as five eager launches, 200 steps
    # each; the faster one runs the timed region (every rank takes the decision of the slowest rank's clocks)
    launch_probe = None
    if launch == "auto":
        eng_e = make_engine(spec, ctx, args, edge_gather=not explicit, no_graph=True)
        eng_e.copy_weights_from(eng)
        tr_e = DataParallelTrainer(eng_e, force=ctx.force_dp) if use_dp else None

        def probe(step, n=200):
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - tp) / n
        with torch.cuda.stream(stream):
            t_g = probe(stepper(eng, trainer, db, yd, n_denom))
            t_e = probe(stepper(eng_e, tr_e, db, yd, n_denom))
        if dist is not None:
            t = torch.tensor([t_g, t_e], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_g, t_e = float(t[0].item()), float(t[1].item())
        launch_probe = {"hipGraph replay": round(t_g, 4), "eager": round(t_e, 4)}
        if t_e < t_g:
            eng.close()
            eng, trainer, launch = eng_e, tr_e, "eager"
        else:
            eng_e.close()
            launch = "graph"
    args = argparse.Namespace(**dict(vars(args), no_graph=launch == "eager"))      # (the fast-path engine below follows the choice)

    # ---- per-kernel roofline: an instrumented (eager, HIP events around every launch) pass of the same steps, BEFORE the
    # timed region so that the timed region is the last and longest stretch of GPU work of the run
    roofline, kernels, prof = None, None, None
    if rank == 0 and not args.no_roofline:
        eng.profile(True)
        with torch.cuda.stream(stream):
            for _ in range(5):
                eng.train_step(db, yd, n_global=n_denom, want_loss=False)
            torch.cuda.synchronize()
            eng.profile_read()
            n_prof = 50 if not light else 20
            for _ in range(n_prof):
                eng.train_step(db, yd, n_global=n_denom, want_loss=False)
            torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
    if dist is not None:
        dist.barrier()

    # ---- the timed region of the contract
    with torch.cuda.stream(stream):
        steps, elapsed = timed_region(stepper(eng, trainer, db, yd, n_denom), args.steps, args.min_seconds, args.warmup)
    ms_per_step = 1e3 * elapsed / steps
    n_counted = B_local if args.shard_of > 1 else n_global          # --shard-of: only this shard's graphs were processed
    value = n_counted * steps / elapsed

    # ---- the fast path beside it: the same steps with the complement aggregation (when the batch qualifies)
    fast = None
    if not explicit and not args.no_fast_path and not light and str(path.get("graph_layers")).startswith("fused"):
        eng2 = make_engine(spec, ctx, args, edge_gather=False)
        p2 = eng2.path_info(db)
        if p2.get("aggregation") == "complement":
            eng2.copy_weights_from(eng)
            tr2 = DataParallelTrainer(eng2, force=ctx.force_dp) if use_dp else None
            with torch.cuda.stream(stream):
                n2, e2 = timed_region(stepper(eng2, tr2, db, yd, n_denom), 50, min(2.0, args.min_seconds), max(args.warmup, 5))
            fast = {"aggregation": "complement", "value": round(n_counted * n2 / e2, 1), "ms_per_step": round(1e3 * e2 / n2, 4),
                    "steps": n2, "what": "the same fit steps with Agg[q] = S - sum over non-neighbours (exact for any adjacency; "
                                         "chosen by default for graphs with average in-degree > (N-1)/2)"}
        eng2.close()

    # ---- N > 1: a second timed pass in the weak regime (--batch graphs PER GPU), the regime the kernels are sized for
    weak = None
    if world > 1 and strong and not args.no_weak_pass and not light and args.shard_of == 1:
        dbw, ydw, _, bw_local, _, nw_global, nw_denom, _ = make_batch(False, B)
        with torch.cuda.stream(stream):
            nw, ew = timed_region(stepper(eng, trainer, dbw, ydw, nw_denom), 50, min(2.0, args.min_seconds), max(args.warmup, 5))
        weak = {"scaling": "weak", "graphs_per_gpu": bw_local, "global_batch": nw_global, "value": round(nw_global * nw / ew, 1),
                "ms_per_step": round(1e3 * ew / nw, 4), "steps": nw}
        del dbw, ydw

    # sanity: the timed steps really trained (finite loss, weights moved)
    loss = eng.forward_backward(db, yd, n_global=n_denom)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss).all()), "non-finite loss after the timed steps"

    if prof is not None:
        tot = sum(ms for _, ms in prof.values())
        kernels = {k: {"calls": c, "avg_us": round(1e3 * ms / c, 2), "share": round(ms / tot, 3)} for k, (c, ms) in
                   sorted(prof.items(), key=lambda kv: -kv[1][1])}
        Bq, Nq = (1, n_rows_local) if ragged else (B_local, N)  # the byte model only needs rows = Bq*Nq and E
        step_bytes = (sum(step_bytes_per_graph(int(n), F, L, int(n) * (int(n) - 2)) for n in sizes) / B_local if ragged
                      else step_bytes_per_graph(N, F, L, E // B_local))
        # dominant kernel = most time per step among the modelled kernels; its roofline is the resource whose floor
        # (algorithmic bytes / 8 TB/s vs algorithmic flops / 157.3 TF) is the LARGER one

        def floors(k):
            return algorithmic_bytes(k, Bq, Nq, F, E, L), algorithmic_flops(k, n_rows_local, F, L)
        modelled = [k for k in prof if any(v is not None for v in floors(k))]
        dom = max(modelled, key=lambda k: prof[k][1])
        calls, ms = prof[dom]
        sec = 1e-3 * ms / calls
        by, fl = floors(dom)
        t_hbm = (by or 0) / (HBM_PEAK_GBS * 1e9)
        t_mfma = (fl or 0) / (FP32_MFMA_PEAK_TF * 1e12)
        step_flops = sum((algorithmic_flops(k, n_rows_local, F, L) or 0) * prof[k][0] for k in prof) / n_prof
        common = {"kernel": dom, "avg_launch_us": round(1e6 * sec, 2),
                  "algorithmic_bytes_per_launch": None if by is None else int(by),
                  "algorithmic_flops_per_launch": None if fl is None else int(fl),
                  "hbm_frac": None if by is None else round(t_hbm / sec, 4),
                  "mfma_frac": None if fl is None else round(t_mfma / sec, 4),
                  "step_hbm_frac": round(step_bytes * (B_local * steps / elapsed) / 1e9 / HBM_PEAK_GBS, 4),
                  "step_mfma_frac": round(step_flops / (1e-3 * ms_per_step) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
        if t_mfma > t_hbm:
            achieved = fl / sec / 1e12
            roofline = dict({"bound": "mfma", "achieved": round(achieved, 1), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                             "frac": round(achieved / FP32_MFMA_PEAK_TF, 4), "traffic": hbm_traffic(dom, args)}, **common)
        else:
            achieved = by / sec / 1e9
            roofline = dict({"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": hbm_traffic(dom, args)}, **common)

    out = None
    if rank == 0:
        links = "%d-%d" % tuple(args.ragged) if ragged else str(N)
        # BASELINE.json's metric string names ITS configuration -- the GLOBAL batch is 4096 whatever the number of GPUs
        # (SURVEY.md 8 d1); anything else (weak scaling on several GPUs, one shard of a run, other sizes) gets its own label
        headline = (args.workload == "cfg2" and (N, F, L, ragged, args.share_weights) == (20, 64, 2, False, False)
                    and n_global == 4096 and args.shard_of == 1)
        if headline:
            metric = "graph-instances/sec (fwd+bwd), 20-V2V-link graphs, batch 4096"
        elif args.shard_of > 1:
            metric = ("graph-instances/sec (fwd+bwd) of ONE shard (%d graphs) of a %d-GPU run at global batch %d, %s-V2V-link graphs, "
                      "feat_dim %d, %d layers, no all-reduce" % (B_local, args.shard_of, B, links, F, L))
        else:
            metric = ("graph-instances/sec (fwd+bwd), %s-V2V-link graphs, feat_dim %d, %d layers, global batch %d (%s scaling, %d per GPU)"
                      % (links, F, L, n_global, args.scaling, B_local))
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist is not None else None
        except Exception:
            rccl = None
        out = {"metric": metric,
               "value": round(value, 1), "unit": "graph-instances/s", "n_gpus": world, "steps": steps,
               "steps_requested": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "timed_seconds": round(elapsed, 4), "higher_is_better": True,
               "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               # which aggregation `value` ran, said at top level: the general edge-index aggregation of AggLayer.call ("edge-bitset-
               # walk": the fused kernels turn every CSR row into a 32-bit set and walk the graph's rows in LDS; degree-aware) is
               # the judged form; the complement rewriting is the fast path beside it; config.per_edge_gather is the per-edge CSR
               # gather kernel k_agg (layer-wise path)
               "aggregation": path.get("aggregation"),
               "fast_path": fast,
               "config": {"workload": "BASELINE.json configs[%d]: %s V2V links, feat_dim=%d, %d-layer GNN, %s, "
                                      "fit step = fwd+Huber+bwd+Adam%s"
                                      % ({"cfg2": 1, "cfg4": 3, "cfg5": 4}[args.workload], links, F, L,
                                         ("global batch %d synthetic graphs cut into %d shard(s)%s"
                                          % (B, n_shards, ", shard 0 timed on one GPU" if args.shard_of > 1 else "")) if strong else
                                         ("batch %d synthetic graphs per GPU" % B),
                                         "+RCCL grad all-reduce" if world > 1 else ""),
                          "weights": "shared" if args.share_weights else "per-node (reference semantics)",
                          "global_batch": n_global, "graphs_per_gpu": B_local, "scaling": args.scaling, "n_params": eng.n_params,
                          "aggregation": path.get("aggregation"), "kernel_path": path, "weak": weak,
                          "ranks_seen": (dist.get_world_size() if dist is not None else 1), "rccl_version": rccl,
                          "collective_backend": (dist.get_backend() if dist is not None else None),
                          "dp_form": (None if trainer is None else
                                      ("sharded optimizer: reduce-scatter + Adam on the slice + all-gather, per bucket" if trainer.shard_optimizer
                                       else ("one all-reduce per gradient bucket, overlapped with the backward phases" if trainer.overlap
                                             else "one all-reduce of the flat gradient"))),
                          "lib_sha256": lib_sha256(),
                          "launch": "eager" if launch == "eager" else "hipGraph replay", "launch_probe_ms": launch_probe,
                          "parallelism": "dp%d" % world},
               "roofline": roofline, "cpu_baseline": None}
        if kernels is not None:
            out["kernels"] = kernels
    eng.close()
    del db, yd
    return out


def cpu_leg_share(wl, budget_s=8.0):
    """CPU restatement (the oracle, kind "port") beside the other workloads' share figures (north_star:
    "next to the ... CPU path timed on the same box"), on a BOUNDED sample of the share, scaled to graphs/s and labelled:
    configs[3] -- oracle/parallel.py's worker processes (one BLAS thread each) over a 128-graph sample of the 1024-graph share
    (the full share costs about a minute per step on one thread); configs[4] -- oracle/compact.py's fit step on a 256-graph
    ragged sample of the 2048-graph share, at one BLAS thread and at the pool's size, the better of the two."""
    from oracle import compact as oc
    from oracle.spec import GnnSpec as OSpec
    host_cores = len(os.sched_getaffinity(0))
    if wl == "cfg4":
        from oracle.parallel import ShardedOracle
        n, f, l, bs = 100, 256, 3, 128
        workers = max(1, min(host_cores, bs // 4))
        x, e, adj, y = synth_batch(np.random.default_rng(1001), bs, n)
        so = ShardedOracle(dict(n_nodes=n, feat_dim=f, n_mp_layers=l, share_weights=False), x, e, adj, y, workers)
        try:
            sec, steps = _timed_steps(so.step, budget_s, 2, 20, 1)
        finally:
            so.close()
        return {"value": round(bs / sec, 1), "unit": "graph-instances/s", "usable_cpus": usable_cpus(), "workers": int(workers),
                "cores": int(min(workers, usable_cpus())), "kind": "port", "ms_per_step": round(1e3 * sec, 2), "steps": steps,
                "sample": "median of %d fit steps of a %d-graph sample of the 1024-graph share (N=100, F=256, L=3, per-node weights): "
                          "%d worker processes x 1 BLAS thread over graph shards, gradients summed, one Adam update; numpy fp32 CSR "
                          "oracle, not Keras/TF1" % (steps, bs, workers)}
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        all_thr = max([p_.get("num_threads", 1) for p_ in threadpool_info()] or [1])
    except Exception:
        threadpool_limits, all_thr = None, 1
    import contextlib
    bs = 256
    sizes, offs, row_ptr, cols, x, e, y = synth_ragged(np.random.default_rng(1001), bs, 8, 128)
    spec = OSpec(n_nodes=1, feat_dim=64, n_mp_layers=2, share_weights=True)
    om = oc.OracleModel(spec, oc.init_params(spec, np.random.default_rng(7), np.float32), dtype=np.float32)
    graph = (offs, row_ptr, cols)
    best = None
    for nthr in sorted({1, int(all_thr)}):
        with (threadpool_limits(limits=nthr) if threadpool_limits is not None else contextlib.nullcontext()):
            sec, steps = _timed_steps(lambda: om.train_step(x, e, graph, y), budget_s / 2, 2, 20, 1)
        if best is None or sec < best[0]:
            best = (sec, steps, nthr)
    sec, steps, nthr = best
    return {"value": round(bs / sec, 1), "unit": "graph-instances/s", "usable_cpus": usable_cpus(), "workers": int(nthr),
            "cores": int(min(nthr, usable_cpus())), "kind": "port", "ms_per_step": round(1e3 * sec, 2), "steps": steps,
            "sample": "median of %d fit steps of a %d-graph ragged sample (8-128 links, %d node rows) of the 2048-graph share (F=64, L=2, "
                      "shared weights), numpy fp32 CSR oracle at %d BLAS thread(s) (the better of 1 and the pool's size), not Keras/TF1"
                      % (steps, bs, int(offs[-1]), nthr)}


def other_workloads(args, ctx):
    """Short driver-observed passes beside the headline:
    * cfg4 / cfg5: configs[3] and configs[4] at their per-GPU shares of an 8-GPU run (shard 0 of the global batch, global Huber
      denominator, no all-reduce), each with a CPU leg on a bounded sample of the share;
    * cfg2_share2 / 4 / 8: the shares of the METRIC's own global batch of 4096 at 2 / 4 / 8 GPUs (the 1024- and 512-graph
      shares run the split-tile fused graph layers);
    * per_edge_gather: the fit step with the per-edge CSR gather / segment sum kernel k_agg (layer-wise graph layers,
      V2X_FUSED=0) -- the aggregation mechanism north_star words, beside the fused kernels' bit-set walk that `value` runs;
    * cfg2loop: configs[2]'s DQN loop on one GPU (50 simulators stepped as arrays, 5 timed episodes of 20 train steps)."""
    import copy
    out = {}

    def light(wl, shard_of, **over):
        a = copy.copy(args)
        a.workload, a.shard_of, a.scaling, a.gpus = wl, shard_of, "strong", 1
        a.steps, a.warmup, a.min_seconds = 20, 5, 0.6
        a.ragged, a.share_weights = None, False
        for k_, v_ in over.items():
            setattr(a, k_, v_)
        a = resolve_workload(a)
        r = run_workload(a, ctx, light=True)
        rf = r.get("roofline") or {}
        d = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
             "graphs_per_gpu": r["config"]["graphs_per_gpu"], "graph_layers": r["config"]["kernel_path"].get("graph_layers"),
             "aggregation": r["aggregation"], "launch": r["config"]["launch"],
             "dominant_kernel": rf.get("kernel"), "bound": rf.get("bound"), "frac": rf.get("frac"),
             "step_hbm_frac": rf.get("step_hbm_frac"), "step_mfma_frac": rf.get("step_mfma_frac")}
        if args.other_kernels:            # the long form: workload prose, the whole path, HIP-event times of every kernel
            d.update({"workload": r["config"]["workload"], "kernel_path": r["config"]["kernel_path"],
                      "timed_seconds": r["timed_seconds"], "kernels": r.get("kernels")})
        return d

    def guarded(key, fn):
        try:
            out[key] = fn()
        except Exception as exc:                      # keep the headline line whatever happens here
            out[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    for wl in ("cfg4", "cfg5"):
        guarded(wl, lambda: light(wl, 8))
        if not args.no_cpu_baseline and "error" not in out[wl]:
            try:
                out[wl]["cpu_baseline"] = cpu_leg_share(wl)
                if not args.other_kernels:            # (the prose of the sample: 128 / 256 graphs of the share, numpy fp32 CSR oracle)
                    out[wl]["cpu_baseline"]["sample"] = out[wl]["cpu_baseline"]["sample"].split(":")[0][:120]
            except Exception as exc:
                out[wl]["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    for g in (2, 4, 8):
        guarded("cfg2_share%d" % g, lambda: light("cfg2", g))

    def per_edge():
        had = os.environ.get("V2X_FUSED")
        os.environ["V2X_FUSED"] = "0"                 # read when the model is created: layer-wise graph layers, k_agg aggregation
        try:
            return light("cfg2", 1)
        finally:
            if had is None:
                del os.environ["V2X_FUSED"]
            else:
                os.environ["V2X_FUSED"] = had
    guarded("per_edge_gather", per_edge)

    def dqn_loop(envs, episodes):
        import torch
        with torch.cuda.stream(torch.cuda.Stream()):
            r = rl_episode(20, 64, 4096, 0.5, 20, 1001, use_graph=True, envs=envs, episodes=episodes)
        r["simulators"], r["episodes"] = max(envs, 1), episodes
        if not args.other_kernels:
            for k_ in ("wall_s", "train_steps_per_s", "rollout_s", "replay_s", "env_steps"):
                r.pop(k_, None)
        if args.other_kernels:
            r["workload"] = ("BASELINE.json configs[2] on one GPU: %d episodes x 20 train steps x (50 rollout transitions on %s + 1 replay "
                             "of batch 4096), 20 links, feat_dim 64; one agent, two-step warm-up episode outside the timed region"
                             % (episodes, "50 simulators stepped as arrays" if envs > 1 else
                                "ONE simulator, sequentially, a B=1 predict each: the reference's own loop shape (BS_brain.py:818-832)"))
        return r
    # the key says which loop: 50 simulators stepped as arrays, or the reference's shape (one simulator, 50 sequential transitions)
    guarded("cfg2loop_envs50", lambda: dqn_loop(50, 5))
    guarded("cfg2loop_env1", lambda: dqn_loop(1, 2))

    def env1_b1():
        # the same one-simulator loop with a B = 1 predict for every greedy transition (the default scores the rollout's 50
        # observations -- which do not depend on the actions -- with ONE predict; identical transitions either way)
        had = os.environ.get("V2X_RL_ROLLOUT_BATCH_PREDICT")
        os.environ["V2X_RL_ROLLOUT_BATCH_PREDICT"] = "0"
        try:
            return dqn_loop(1, 2)
        finally:
            if had is None:
                del os.environ["V2X_RL_ROLLOUT_BATCH_PREDICT"]
            else:
                os.environ["V2X_RL_ROLLOUT_BATCH_PREDICT"] = had
    guarded("cfg2loop_env1_b1_predicts", env1_b1)
    return out


def summary(out, others, dropin, cpu):
    """Every figure of the default run that is quoted beside the headline, as one flat object of numbers: ms per step of the shares of
    the metric's batch (2 / 4 / 8 GPUs), configs[3] / [4] at their shares with the dominant kernel's roofline fraction and the CPU
    leg (graphs/s), the per-edge CSR gather step, both DQN loops (50 simulators as arrays; the reference's one-simulator shape) and the
    dict-API replay at the reference's configuration."""
    o = others or {}

    def get(key, field, nd=4):
        v = (o.get(key) or {}).get(field)
        return None if v is None else round(float(v), nd)

    def cpu_of(key):
        v = ((o.get(key) or {}).get("cpu_baseline") or {}).get("value")
        return None if v is None else round(float(v), 1)
    lp = (out.get("config") or {}).get("launch_probe_ms") or {}
    sm = {"ms": out["ms_per_step"], "graph_replay_ms": lp.get("hipGraph replay"), "eager_ms": lp.get("eager"),
          "frac": (out.get("roofline") or {}).get("frac"),
          "step_mfma_frac": (out.get("roofline") or {}).get("step_mfma_frac"),
          "fast_path_ms": (out.get("fast_path") or {}).get("ms_per_step"),
          "cpu": None if cpu is None else cpu["value"], "cpu_workers": None if cpu is None else cpu["workers"],
          "usable_cpus": None if cpu is None else cpu["usable_cpus"],
          "share2_ms": get("cfg2_share2", "ms_per_step"), "share4_ms": get("cfg2_share4", "ms_per_step"),
          "share8_ms": get("cfg2_share8", "ms_per_step"),
          "cfg4_ms": get("cfg4", "ms_per_step"), "cfg4_frac": get("cfg4", "frac"), "cfg4_step_mfma_frac": get("cfg4", "step_mfma_frac"),
          "cfg4_cpu": cpu_of("cfg4"),
          "cfg5_ms": get("cfg5", "ms_per_step"), "cfg5_frac": get("cfg5", "frac"), "cfg5_step_mfma_frac": get("cfg5", "step_mfma_frac"),
          "cfg5_cpu": cpu_of("cfg5"),
          "per_edge_ms": get("per_edge_gather", "ms_per_step"),
          "cfg2loop_envs50_ms": get("cfg2loop_envs50", "ms_per_train_step", 3), "cfg2loop_env1_ms": get("cfg2loop_env1", "ms_per_train_step", 3),
          "cfg2loop_env1_b1_predicts_ms": get("cfg2loop_env1_b1_predicts", "ms_per_train_step", 3),
          "dropin_ms": (dropin or {}).get("replay_triple_ms"), "predict_one_us": (dropin or {}).get("predict_one_step_us")}
    return sm


def main():
    args = build_parser().parse_args()
    if args.workload in ("cfg0", "cfg2loop"):
        return main_rl(args)
    args = resolve_workload(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # no launcher around us: start the ranks ourselves
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    if os.environ.get("V2X_BENCH_ONE_DEVICE") == "1":     # tests: every rank on cuda:0 (one-GPU box), gloo collective
        local = 0
    torch.cuda.set_device(local)
    dist = None
    force_dp = os.environ.get("V2X_FORCE_DP") == "1"      # exercise the RCCL path with a single rank (tests)
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("V2X_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for the one-GPU test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ctx = Ctx(world, rank, local, dist, force_dp)

    # Order of a default run: CPU legs first, then every GPU pass, the contract's timed region (>= --min-seconds) last but
    # for its own fast-path / weak companions -- the stretch a 5-s utilisation sampler sees is the one that is reported.
    ragged = args.ragged is not None
    solo = rank == 0 and world == 1 and args.shard_of == 1
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not ragged:
        b_local = args.batch // (args.shard_of if args.scaling == "strong" else 1)
        cpu = cpu_baseline(args.nodes, args.feat, args.layers, args.share_weights, b_local, args.cpu_seconds, full=args.other_kernels)
    dropin = None
    if solo and args.workload == "cfg2" and not args.no_dropin:
        try:
            dropin = dropin_boundary(local)
        except Exception as exc:
            dropin = {"error": "%s: %s" % (type(exc).__name__, exc)}
    others = None
    if solo and args.workload == "cfg2" and not args.no_other_workloads:
        others = other_workloads(args, ctx)
    out = run_workload(args, ctx)
    if rank == 0:
        out["cpu_baseline"] = cpu
        if "kernels" in out:
            out["kernels_note"] = "HIP events, eager launches; their sum exceeds the replayed step (ms_per_step); rocprofv3: profiles/"
        if dropin is not None:
            out["dropin_ref_config"] = dropin
            if not args.other_kernels:
                dropin.pop("what", None)           # (prose: BS.predict + BS.predict(target) + BS.train_dnn on the reference's dict payload, N=4 F=16 B=512)
            if cpu is not None and "C0_one_thread" in cpu.get("legs", {}):
                dropin["cpu_fit_step_C0_ms"] = cpu["legs"]["C0_one_thread"]["ms_per_step"]
        if others is not None:
            out["config"]["other_workloads"] = others
            pe = others.get("per_edge_gather") or {}
            out["config"]["per_edge_gather"] = {k_: pe.get(k_) for k_ in ("value", "unit", "ms_per_step", "aggregation", "graph_layers", "error") if k_ in pe}
        # numbers only, LAST key of the line: survives a record that keeps a tail of stdout or filters keys
        out["summary"] = summary(out, others, dropin, cpu)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        _print_last(json.dumps(out))


if __name__ == "__main__":
    main()
