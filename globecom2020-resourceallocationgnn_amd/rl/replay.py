"""Device-resident replay memory: the transitions `Agent.train_observe` stores (BS_brain.py:245-270, :552) kept in
HBM in the engine's packed layout, so that a replay step (BS_brain.py:555-748) is gather -> online forward -> target
forward -> target rule -> fit without the minibatch ever visiting the host.

torch tensors own the memory; every data movement on the device is one of the C-ABI kernels
`v2x_gather_rows` / `v2x_dqn_targets` (include/v2xgnn.h).  Sampling indices still come from the caller's numpy RNG
(same draws as `Memory.sample`), 4 bytes per sampled transition.

Layout per transition (n links):  xe[n][16] of s, xe'[n][16] of s', the adjacency as one source bit mask per link
(mask[n] int32, bit p of mask[q] = edge p -> q), action[n] int32, reward double -- and, for the regular case, the
ready-made CSR columns col_idx[E].  The reference topology gives every link the in-degree n-2 (self and own receiver
excluded, BS_brain.py:441-445), so a gathered minibatch is CSR with a constant row pointer and the gathered col_idx
rows as is.  The simulator can, rarely, make a link its own receiver (two vehicles on the same spot: the distance sort
of Environment.py:365-375 then puts the other one first); such transitions have in-degree n-1 and a minibatch that
contains one is expanded from the masks instead (torch ops, variable edge count).  New transitions are staged on the host and flushed in one copy per tensor before
the next sample (the reference stores 50 transitions between replays, BS_brain.py:758).
"""
import ctypes as C
import os

import numpy as np

from .. import lib as _lib
from ..engine import DeviceBatch, current_stream_ptr
from ..packing import adj_to_csr, pack_xe


class DeviceReplay(object):
    MAX_LINKS = 31          # one int32 source mask per link (bit p of mask[q]: p sends to q)

    def __init__(self, capacity, n_nodes, device=0):
        import torch
        self.torch = torch
        self.capacity, self.n = int(capacity), int(n_nodes)
        self.device = torch.device("cuda:%d" % device) if isinstance(device, int) else torch.device(device)
        self._lib = _lib.load_library()
        self.size, self.head = 0, 0            # stored transitions; slot the next transition goes to (ring)
        self.n_edges = None
        self._alloc = 0
        self._stage = []
        self._row_ptr = {}
        self._bufs = {}
        self._idx_ring = {}
        self._n_staged = 0
        self._regular = np.ones(0, bool)       # host-side: slot holds a graph with in-degree n-2 everywhere
        self._early = None                     # stage_early(): (slot, xe, col, mask) already copied for the next packed block
        self._db_cache = {}

    # ------------------------------------------------------------------ storage
    def _grow(self, need):
        """Storage grows geometrically up to `capacity` (the reference's 1e6-transition cap is 4.1 GB at 20 links).  The
        first allocation holds 65,536 transitions (275 MB at 20 links, 0.1 % of the HBM): growing means new tensors and a
        copy, 45-50 ms on the host -- paid at 65,536, 131,072, ... stored transitions instead of inside the first hundred
        train steps (tools/prof_rl_sections.py)."""
        if need <= self._alloc:
            return
        torch, n, E = self.torch, self.n, self.n_edges
        new = min(self.capacity, max(need, 2 * self._alloc, 65536))
        def grow(old, shape, dtype):
            t = torch.zeros((new,) + shape, dtype=dtype, device=self.device)
            if old is not None:
                t[:old.shape[0]] = old
            return t
        first = self._alloc == 0
        self.xe = grow(None if first else self.xe, (n, 16), torch.float32)
        self.xe_next = grow(None if first else self.xe_next, (n, 16), torch.float32)
        self.col = grow(None if first else self.col, (max(E, 1),), torch.int32)
        self.mask = grow(None if first else self.mask, (n,), torch.int32)
        reg = np.ones(new, bool)
        reg[:self._alloc] = self._regular[:self._alloc]
        self._regular = reg
        self.action = grow(None if first else self.action, (n,), torch.int32)
        self.reward = grow(None if first else self.reward, (), torch.float64)
        self._alloc = new

    def add(self, x, e, adj, action, reward, x_next, e_next):
        """x, x_next [n, Dn]; e, e_next [n, De]; adj [n, n] (Adj[p, q] = 1: p sends to q); action [n]; reward scalar."""
        adj = np.asarray(adj)
        if self.n > self.MAX_LINKS:
            raise ValueError("DeviceReplay keeps the adjacency as one 32-bit source mask per link: at most 31 links")
        row_ptr, col, _ = adj_to_csr(adj[None])
        if self.n_edges is None:
            self.n_edges = self.n * (self.n - 2)
        regular = col.shape[0] == self.n_edges and not np.any(np.diff(row_ptr) != self.n - 2)
        mask = ((adj != 0).astype(np.int64) << np.arange(self.n, dtype=np.int64)[:, None]).sum(axis=0).astype(np.int32)   # [q]: bits p
        colrow = col.astype(np.int32) if regular else np.zeros(max(self.n_edges, 1), np.int32)
        self._stage.append((pack_xe(np.asarray(x, np.float32), np.asarray(e, np.float32))[None],
                            pack_xe(np.asarray(x_next, np.float32), np.asarray(e_next, np.float32))[None],
                            colrow[None], np.asarray(action, np.int32).reshape(1, -1), np.array([float(reward)], np.float64),
                            mask[None], np.array([regular], bool)))
        self._n_staged += 1

    def add_many(self, x, e, adj, action, reward, x_next, e_next):
        """K transitions at once (the batched rollout stores one per environment and step): x, x_next [K, n, Dn]; e, e_next
        [K, n, De]; adj [K, n, n]; action [K, n]; reward [K].  Same staged tuples as K calls of add(), built with array
        operations."""
        adj = np.asarray(adj)
        K, n = adj.shape[0], self.n
        if n > self.MAX_LINKS:
            raise ValueError("DeviceReplay keeps the adjacency as one 32-bit source mask per link: at most 31 links")
        if self.n_edges is None:
            self.n_edges = n * (n - 2)
        nz = adj != 0
        deg = nz.sum(axis=1)                                             # [K, q] in-degrees
        regular = np.all(deg == n - 2, axis=1)
        mask = (nz.astype(np.int64) << np.arange(n, dtype=np.int64)[None, :, None]).sum(axis=1).astype(np.int32)   # [K, q]: bits p
        col = np.zeros((K, max(self.n_edges, 1)), np.int32)
        if regular.any():                                               # CSR by destination, ascending sources (adj_to_csr order)
            src = np.nonzero(np.transpose(nz[regular], (0, 2, 1)))[2].astype(np.int32)
            col[regular] = src.reshape(int(regular.sum()), self.n_edges)
        x, e = np.asarray(x, np.float32), np.asarray(e, np.float32)
        xn, en = np.asarray(x_next, np.float32), np.asarray(e_next, np.float32)
        xe = pack_xe(x.reshape(K * n, -1), e.reshape(K * n, -1)).reshape(K, n, -1)
        xe_next = pack_xe(xn.reshape(K * n, -1), en.reshape(K * n, -1)).reshape(K, n, -1)
        action = np.asarray(action, np.int32).reshape(K, n)
        reward = np.asarray(reward, np.float64).reshape(K)
        self._stage.append((xe, xe_next, col, action, reward, mask, regular))      # one block of K transitions
        self._n_staged += K

    def add_many_packed(self, xe, xe_next, col, mask, regular, action, reward):
        """add_many for observations that are ALREADY in the stored layout (BatchedEnviron.observe_packed: xe / xe_next
        [K, n, 16] float32, col [K, n (n-2)] int32, mask [K, n] int32, regular [K] bool): staged as they are.  The caller
        must not write to the arrays afterwards (the simulator hands out fresh ones every step)."""
        K, n = mask.shape
        if n != self.n or n > self.MAX_LINKS:
            raise ValueError("add_many_packed: %d links, this memory holds %d (at most %d)" % (n, self.n, self.MAX_LINKS))
        if self.n_edges is None:
            self.n_edges = n * (n - 2)
        if (xe.dtype != np.float32 or xe.shape != (K, n, 16) or xe_next.shape != xe.shape or xe_next.dtype != np.float32
                or col.dtype != np.int32 or col.shape != (K, max(self.n_edges, 1)) or mask.dtype != np.int32):
            raise ValueError("add_many_packed: arrays are not in the packed layout")
        early = self._early
        if early is not None and not (not self._stage and early[1] is xe and early[2] is col and early[3] is mask):
            self._early = None                 # something else was staged in between: the plain path (the rows are rewritten)
        self._stage.append((xe, xe_next, col, np.asarray(action, np.int32).reshape(K, n), np.asarray(reward, np.float64).reshape(K),
                            mask, np.asarray(regular, bool)))
        self._n_staged += K

    def stage_early(self, xe, col, mask):
        """The observation half of the block the NEXT add_many_packed will stage (same array objects), copied to its slots
        now -- the caller has time to spare before it knows the actions (the GPU is still fitting); flush() then copies
        only xe_next / action / reward.  Not taken (False) when anything is staged already or the ring is about to wrap."""
        K = xe.shape[0]
        if self.n_edges is None:
            self.n_edges = self.n * (self.n - 2)
        if (self._stage or self._early is not None or self.device.type != "cuda" or K > 4096
                or self.size + K > self.capacity or self.head + K > self.capacity or self.head != self.size):
            return False
        self._grow(self.size + K)
        pos = self.head
        for t, a, name in ((self.xe, xe, "xe"), (self.col, col, "col"), (self.mask, mask, "mask")):
            self._to_device(t[pos:pos + K], a, name)
        self._early = (pos, xe, col, mask)
        return True

    def prefetch_indices(self, idx, K):
        """Upload the slots of a minibatch drawn AHEAD: idx are positions in the FIFO order as it will be once K more
        transitions are stored.  -> what sample(idx, pre=...) takes, or None when the ring wraps by then."""
        if self.size + self._n_staged + K > self.capacity:
            return None
        slots = np.asarray(idx, np.int64).astype(np.int32)               # not wrapped: FIFO position = slot
        return slots, self._upload_indices(slots)

    def __len__(self):
        return min(self.capacity, self.size + self._n_staged)

    def _to_device(self, dst, src, name):
        """dst (a slice of a storage tensor) <- src (host array) through a pinned staging buffer: the copy is asynchronous, the
        buffer is one of four per (tensor, shape) and its event guards the reuse.  (Measured and dropped: v2x_gather_rows reading the
        pinned buffer directly as the copy engine -- no faster on the host, slower on the device.)"""
        torch = self.torch
        key = ("stage", name, tuple(src.shape))
        ring = self._idx_ring.get(key)
        if ring is None:
            ring = self._idx_ring[key] = {"pin": [torch.empty(tuple(src.shape), dtype=dst.dtype).pin_memory() for _ in range(4)],
                                          "ev": [None] * 4, "next": 0}
            ring["np"] = [t.numpy() for t in ring["pin"]]
        i = ring["next"]
        ring["next"] = (i + 1) % 4
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()
        np.copyto(ring["np"][i], src)
        dst.copy_(ring["pin"][i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["ev"][i] = ev

    def flush(self):
        """Staged transitions -> HBM (one copy per tensor; ring wrap handled by splitting at the end of the storage)."""
        if not self._stage:
            return
        torch = self.torch
        k = self._n_staged
        self._grow(min(self.capacity, self.size + k))
        one = len(self._stage) == 1                  # the batched rollout stages one block per train step: no re-stacking
        cols = [self._stage[0][i] if one else np.concatenate([s[i] for s in self._stage]) for i in range(6)]
        regular = self._stage[0][6] if one else np.concatenate([s[6] for s in self._stage])
        dst = (self.xe, self.xe_next, self.col, self.action, self.reward, self.mask)
        pos, done = self.head, 0
        skip = self._early is not None and one and self._early[0] == pos and self._early[1] is cols[0]
        self._early = None
        while done < k:
            m = min(k - done, self._alloc - pos)
            for t, a, name in zip(dst, cols, ("xe", "xe_next", "col", "action", "reward", "mask")):
                if skip and name in ("xe", "col", "mask"):
                    continue                           # stage_early() copied them
                if self.device.type == "cuda" and m <= 4096:
                    self._to_device(t[pos:pos + m], a[done:done + m], name)
                else:
                    t[pos:pos + m].copy_(torch.from_numpy(np.ascontiguousarray(a[done:done + m])))
            self._regular[pos:pos + m] = regular[done:done + m]
            pos, done = pos + m, done + m
            if pos == self.capacity:                   # full ring: overwrite the oldest transitions
                pos = 0
        self.head = pos
        self.size = min(self.capacity, self.size + k)
        self._stage = []
        self._n_staged = 0

    # ------------------------------------------------------------------ sampling
    def _gather(self, src, idx_dev, k, name):
        """Gather into a buffer that is REUSED for every minibatch of k transitions: stable device pointers let the
        engine replay its captured hipGraph instead of re-capturing one per replay step."""
        torch = self.torch
        key = (name, k)
        if key not in self._bufs:
            self._bufs[key] = torch.empty((k,) + tuple(src.shape[1:]), dtype=src.dtype, device=self.device)
        out = self._bufs[key]
        row_bytes = src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
        rc = self._lib.v2x_gather_rows(src.data_ptr(), idx_dev.data_ptr(), out.data_ptr(), k, row_bytes,
                                       current_stream_ptr(self.device.index))
        _lib.check(self._lib, rc, None)
        return out

    def _gather_many(self, jobs, idx_dev, k):
        """_gather of several storage tensors by the same index list as ONE launch (v2x_gather_rows_multi): the five gathers
        of a minibatch were five launches and five library calls per replay step."""
        torch = self.torch
        outs = []
        n = len(jobs)
        key = ("jobs", k, tuple(name for _, name in jobs))
        tab = self._bufs.get(key)
        if tab is None or any(tab[3][j] != jobs[j][0].data_ptr() for j in range(n)):      # (storage re-allocated by _grow: new table)
            src_p, dst_p, rb = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
            for j, (src, name) in enumerate(jobs):
                bk = (name, k)
                if bk not in self._bufs:
                    self._bufs[bk] = torch.empty((k,) + tuple(src.shape[1:]), dtype=src.dtype, device=self.device)
                src_p[j], dst_p[j] = src.data_ptr(), self._bufs[bk].data_ptr()
                rb[j] = src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
            tab = self._bufs[key] = (src_p, dst_p, rb, [src.data_ptr() for src, _ in jobs])
        rc = self._lib.v2x_gather_rows_multi(n, tab[0], tab[1], tab[2], idx_dev.data_ptr(), k, current_stream_ptr(self.device.index))
        _lib.check(self._lib, rc, None)
        for _, name in jobs:
            outs.append(self._bufs[(name, k)])
        return outs

    def q_stats(self, y, k, n_channels):
        """[2, n] float64 device tensor: per link the SUM of all entries of the k fitted targets y [k * n, C] and the sum of their
        per-sample maxima (BS_brain.py:743-746 divides by k * C and by k) -- one launch (v2x_q_stats)."""
        # every call gets a row of its own (a deferred caller reads a whole episode's rows at the end): rows are never reused, an
        # exhausted block is simply replaced -- views of the old one keep it alive
        blk = self._bufs.get(('qstats',))
        if blk is None or blk[1] == blk[0].shape[0]:
            blk = self._bufs[('qstats',)] = [self.torch.empty((64, 2, self.n), dtype=self.torch.float64, device=self.device), 0]
        out = blk[0][blk[1]]
        blk[1] += 1
        rc = self._lib.v2x_q_stats(y.data_ptr(), k, self.n, n_channels, out.data_ptr(), current_stream_ptr(self.device.index))
        _lib.check(self._lib, rc, None)
        return out

    def _upload_indices(self, slots):
        """The storage slots of a minibatch where the gather kernels can read them: one of four PINNED host buffers, which the
        device addresses directly (unified addressing) -- 16 KB read over the bus by the kernels themselves instead of a copy
        launch of ours.  The buffer's event (recorded by sample() behind its last gather) guards the reuse."""
        torch, k = self.torch, len(slots)
        ring = self._idx_ring.get(k)
        if ring is None:
            ring = self._idx_ring[k] = {"pin": [torch.empty(k, dtype=torch.int32).pin_memory() for _ in range(4)],
                                        "ev": [None] * 4, "next": 0, "dev": None}
            # zero copy needs the pinned buffers mapped at their own address (checked, not assumed); V2X_RL_ZERO_COPY=0, or a
            # platform without unified addressing: a copy launch into a device buffer of the same ring instead
            zero_copy = os.environ.get("V2X_RL_ZERO_COPY", "1") != "0" and all(
                self._lib.v2x_device_addressable(t.data_ptr()) == 1 for t in ring["pin"])
            if not zero_copy:
                ring["dev"] = [torch.empty(k, dtype=torch.int32, device=self.device) for _ in range(4)]
        i = ring["next"]
        ring["next"] = (i + 1) % 4
        if ring["ev"][i] is not None:
            ring["ev"][i].synchronize()
            ring["ev"][i] = None
        ring["pin"][i].numpy()[:] = slots
        self._idx_in_use = (ring, i)
        if ring["dev"] is not None:
            ring["dev"][i].copy_(ring["pin"][i], non_blocking=True)
            return ring["dev"][i]
        return ring["pin"][i]

    def _indices_consumed(self):
        """call behind the last kernel that reads the buffer _upload_indices handed out"""
        use, self._idx_in_use = getattr(self, "_idx_in_use", None), None
        if use is not None:
            ev = self.torch.cuda.Event()
            ev.record()
            use[0]["ev"][use[1]] = ev

    def row_ptr(self, k):
        if k not in self._row_ptr:
            deg = self.n_edges // self.n
            self._row_ptr[k] = (self.torch.arange(k * self.n + 1, dtype=self.torch.int32, device=self.device) * deg)
        return self._row_ptr[k]

    def logical_to_slot(self, idx):
        """Index into the FIFO order (0 = oldest stored transition, as in `Memory.samples`) -> storage slot."""
        idx = np.asarray(idx, np.int64)
        start = self.head if self.size == self.capacity else 0
        return ((start + idx) % self.capacity).astype(np.int32)

    def sample(self, idx, pre=None):
        """-> (batch of s, batch of s', action [k, n] int32, reward [k] float64), all in HBM.
        pre: prefetch_indices(idx, ...) of the same idx (their slots are on the device already)."""
        self.flush()
        torch = self.torch
        k = len(idx)
        if pre is not None:
            slots, idx_dev = pre
        else:
            slots = self.logical_to_slot(idx)
            idx_dev = self._upload_indices(slots)
        regular = bool(self._regular[slots].all())
        xe, xe_next, action, reward, last = self._gather_many(
            [(self.xe, 'xe'), (self.xe_next, 'xe_next'), (self.action, 'action'), (self.reward, 'reward'),
             (self.col, 'col') if regular else (self.mask, 'mask')], idx_dev, k)
        xe, xe_next = xe.view(k * self.n, 16), xe_next.view(k * self.n, 16)
        if regular:                                                # every sampled graph has in-degree n-2: CSR as stored
            col = last.view(-1)
            self._indices_consumed()
            rp, max_edges = self.row_ptr(k), self.n_edges
            dbs = self._db_cache.get(k)                # the gather buffers are reused: so are the batch descriptors
            if dbs is None or dbs[0].xe.data_ptr() != xe.data_ptr() or dbs[0].col_idx.data_ptr() != col.data_ptr():
                dbs = self._db_cache[k] = tuple(DeviceBatch.from_tensors(k, self.n, t, rp, col, max_edges) for t in (xe, xe_next))
            return dbs[0], dbs[1], action, reward
        else:                                                      # expand the source masks (ascending sources per row)
            masks = last                                                                          # [k, n(q)]
            self._indices_consumed()
            bits = ((masks[:, :, None] >> torch.arange(self.n, device=self.device, dtype=torch.int32)) & 1).bool()
            col = bits.nonzero()[:, 2].to(torch.int32).contiguous()                               # (graph, q, p) order
            rp = torch.zeros(k * self.n + 1, dtype=torch.int32, device=self.device)
            rp[1:] = torch.cumsum(bits.sum(dim=2).view(-1), 0).to(torch.int32)
            max_edges = self.n * (self.n - 1)
        mk = lambda t: DeviceBatch.from_tensors(k, self.n, t, rp, col, max_edges)
        return mk(xe), mk(xe_next), action, reward

    def target_buffer(self, k, n_channels):
        """Reused [k * n, C] buffer for the training targets of a k-transition minibatch."""
        if ('y', k) not in self._bufs:
            self._bufs[('y', k)] = self.torch.empty((k * self.n, n_channels), dtype=self.torch.float32, device=self.device)
        return self._bufs[('y', k)]

    def dqn_targets(self, q, q_next, action, reward, gamma):
        """The target rule (BS_brain.py:684-692) on device: y = q with y[b, k, a[b, k]] = r[b] + gamma * max q'[b, k]."""
        k, n, Cc = action.shape[0], self.n, q.shape[1]
        if ('y', k) not in self._bufs:
            self._bufs[('y', k)] = self.torch.empty_like(q)
        y = self._bufs[('y', k)]
        rc = self._lib.v2x_dqn_targets(q.data_ptr(), q_next.data_ptr(), action.data_ptr(), reward.data_ptr(),
                                       C.c_double(float(gamma)), k, n, Cc, y.data_ptr(),
                                       current_stream_ptr(self.device.index))
        _lib.check(self._lib, rc, None)
        return y
