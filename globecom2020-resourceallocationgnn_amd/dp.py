"""Batched data parallelism over independent graph instances.

One process per GPU; every rank holds a full replica of the weights and Adam state and a
contiguous shard of B/G whole graphs of the replay minibatch (AggLayer only contracts inside a
sample, BS_brain.py:73, so forward and backward need no communication).  The only collective
is ONE all-reduce (sum) of the flat fp32 gradient buffer per step -- RCCL over xGMI through
`torch.distributed` (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).  Each rank
differentiates its shard of the GLOBAL Huber mean (n_global = B), so the sum of the per-rank
gradients is exactly the single-GPU gradient and no rescale is needed.  The target-network
sync stays a local device-to-device copy.
"""
import numpy as np


class DataParallelTrainer(object):
    """`backend` implements forward_backward(batch, y, n_global, want_loss) -> loss tensor,
    grad_tensor() -> flat torch tensor aliasing the gradient buffer, apply_gradients().
    `GnnEngine` is the GPU backend."""

    def __init__(self, backend, process_group=None, force=False, overlap=None, shard_optimizer=None):
        import torch.distributed as dist
        self.dist = dist
        self.backend = backend
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self._grad = None
        self._buckets = None
        self.force = force          # run the collective even with one rank (exercises the RCCL path)
        # overlap: split the step so that the Dense-layer bucket is all-reduced during the graph-layer backward
        # (train_step).  None = the V2X_DP_OVERLAP environment switch, default OFF: the split costs two launches of the
        # slab sums instead of one and two stream hand-overs per step; measured on one MI355X with the RCCL path forced
        # (tools/dp_host_overhead.py, 20 links x 64 features, batch 4096, 3 MB of gradients): 301 us per step split
        # against 271 us unsplit (264 without data parallelism), i.e. the split only pays once hiding the Dense bucket's
        # all-reduce (1.2 MB, latency-bound on xGMI) is worth more than 30 us -- and the models it applies to (feature
        # width <= 64) have at most a few MB of gradients.
        import os
        # Wide models (feat_dim >= 128: one bucket per layer, configs[3]: 207.6 MB of gradients) overlap by default: measured on
        # one MI355X with the RCCL path forced (tools/dp_host_overhead.py --wide) the phased step costs 3.82 ms against 3.65 ms
        # with one all-reduce after the merged weight-gradient launch -- 0.17 ms, which the all-reduce of four of the five
        # buckets behind the remaining backward repays as soon as the collective takes longer than that (8 GPUs over xGMI:
        # ~1 ms for 207.6 MB; unmeasured here, one GPU per box).
        wide = getattr(getattr(backend, "spec", None), "feat_dim", 0) >= 128
        env = os.environ.get("V2X_DP_OVERLAP")
        self.overlap = ((env == "1") if env is not None else wide) if overlap is None else bool(overlap)
        # shard_optimizer: reduce-scatter + Adam on the rank's 1 / G slice of every bucket + all-gather of the parameters
        # (V2X_DP_SHARD_OPTIMIZER=1); see train_step
        self.shard_optimizer = (os.environ.get("V2X_DP_SHARD_OPTIMIZER", "0") == "1") if shard_optimizer is None else bool(shard_optimizer)
        self._has_reduce_scatter = True
        self._bucket_ranges = None
        self._param = None

    def shard(self, batch, y):
        """Contiguous shard of whole graphs for this rank (variable-size batches: balanced by edges + nodes,
        PackedBatch.shard_bounds).  y is [R, C] for the full batch."""
        if self.world == 1:
            return batch, y
        sh, (r0, r1) = batch.shard(self.rank, self.world, with_rows=True)
        return sh, y[r0:r1]

    def _overlapped(self, local_batch):
        """Bucketed collectives overlapped with the backward pass: available when the backend splits its step
        (GnnEngine.forward_backward_phase) and the batch is resident on the device."""
        return ((self.overlap or self.shard_optimizer) and hasattr(self.backend, "forward_backward_phase")
                and (hasattr(local_batch, "device") or getattr(self.backend, "phases_on_host", False)))

    # ---- collectives on one bucket -------------------------------------------------------------------------------------
    def _slice_of(self, n):
        """(first, count) of this rank's slice of an n-float bucket for reduce-scatter / sharded Adam, or None when the bucket
        does not cut into `world` float4-aligned equal slices (it is then all-reduced and updated whole)."""
        if n % (4 * self.world):
            return None
        c = n // self.world
        return self.rank * c, c

    def _reduce_scatter(self, bucket, first, count):
        """Sum over ranks of `bucket`, this rank's slice only, left in place in bucket[first : first + count].  RCCL: a real
        reduce-scatter (half the bytes of an all-reduce); backends without one (gloo in the CPU tests): all-reduce."""
        out = bucket[first:first + count]
        if self._has_reduce_scatter:
            try:
                return self.dist.reduce_scatter_tensor(out, bucket, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            except (RuntimeError, NotImplementedError):
                self._has_reduce_scatter = False
        return self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def train_step(self, local_batch, local_y, n_graphs_global=None, want_loss=True, n_denominator=None):
        """forward+backward on the local shard, reduce the gradient over the ranks, Adam.

        n_denominator (n_graphs_global is its older name): what the GLOBAL Huber mean divides by besides the channel
        count -- the number of graphs of the whole minibatch for fixed-size graphs (one mean per output, BS_brain.py:214),
        the number of NODE ROWS of the whole minibatch for variable_graphs models (one mean over all rows): ragged shards
        hold different numbers of rows, so the caller sums them over the ranks (bench.py does).

        Three forms:
        * default: one replayed graph for forward + backward + slab sums, ONE all-reduce of the flat gradient, Adam on every
          rank -- right for the headline model (3 MB of gradients: latency-bound on xGMI, nothing to hide);
        * `overlap`: the step runs in phases (GnnEngine.forward_backward_phase); bucket k of the gradient is final after phase
          k and its all-reduce is started right away -- asynchronously, RCCL works on its own stream -- while the later
          phases compute.  Wide models have one bucket per layer (configs[3]: 4.6 + 3 x 13.5 + 6.9 M floats = 207.6 MB, whose
          all-reduce is bandwidth-bound and worth hiding behind the ~2.4 ms of backward that follow the first bucket);
          narrow models have two (measured slower than the default at 3 MB: off by default);
        * `shard_optimizer` (implies the phases): reduce-scatter instead of all-reduce, Adam on this rank's 1 / G slice of
          every bucket (v2x_apply_gradients_range: the 28 bytes per parameter of the Adam pass shrink by G), all-gather of
          the updated parameters.  Optimizer moments are then only valid on their owner's slice (gather_optimizer_state)."""
        if n_denominator is not None:
            n_graphs_global = n_denominator
        if n_graphs_global is None:
            raise ValueError("train_step: the global Huber denominator (n_denominator) is required under data parallelism")
        reduce_now = self.world > 1 or self.force
        if reduce_now and self._overlapped(local_batch):
            return self._train_step_phased(local_batch, local_y, n_graphs_global, want_loss)
        if reduce_now and self.shard_optimizer:
            # The phases are not available for this batch (a host-resident batch, a backend without them), but the optimizer
            # state is SHARDED: after any sharded step a rank holds current Adam moments for its own slices only, so a step
            # that ran full-range Adam on every rank would apply different moments on different ranks and the replicas
            # would part without an error.  Stay sharded: whole backward first, then the same per-bucket
            # reduce-scatter / Adam-on-the-slice / all-gather.
            if not hasattr(self.backend, "apply_gradients_range") or not hasattr(self.backend, "grad_buckets"):
                raise RuntimeError("shard_optimizer: the backend has neither phases nor apply_gradients_range; a full-range Adam "
                                   "step would desynchronise the sharded optimizer state")
            return self._train_step_phased(local_batch, local_y, n_graphs_global, want_loss, phased=False)
        loss = self.backend.forward_backward(local_batch, local_y, n_global=n_graphs_global, want_loss=want_loss)
        if reduce_now:
            if self._grad is None:
                self._grad = self.backend.grad_tensor()
            self.dist.all_reduce(self._grad, op=self.dist.ReduceOp.SUM, group=self.group)
            if want_loss and loss is not None:
                loss = self._reduce_loss(loss)
        self.backend.apply_gradients()
        return loss

    def _reduce_loss(self, loss):
        import torch
        if not torch.is_tensor(loss):
            # host-side loss (numpy batches): reduce it on the device the gradient lives on (RCCL has no
            # CPU tensors; gloo takes the CPU tensor as is)
            loss_t = torch.as_tensor(np.asarray(loss, np.float64), device=self._grad.device)
            self.dist.all_reduce(loss_t, op=self.dist.ReduceOp.SUM, group=self.group)
            return loss_t.cpu().numpy()
        self.dist.all_reduce(loss, op=self.dist.ReduceOp.SUM, group=self.group)
        return loss

    def _train_step_phased(self, local_batch, local_y, n_global, want_loss, phased=True):
        """phased=False: the backend's unsplit forward + backward first, then the bucket collectives (nothing overlaps; the
        optimizer step is the sharded one all the same)."""
        if self._grad is None:
            self._grad = self.backend.grad_tensor()
        if self._buckets is None:
            self._bucket_ranges = list(self.backend.grad_buckets())
            self._buckets = [self._grad[o:o + n] for o, n in self._bucket_ranges]
            if self.shard_optimizer:
                self._param = self.backend.param_tensor()
        nb = len(self._buckets)
        works, loss = [], None
        if not phased:
            loss = self.backend.forward_backward(local_batch, local_y, n_global=n_global, want_loss=want_loss)
        for k in range(nb):
            if phased:
                loss = self.backend.forward_backward_phase(local_batch, local_y, k, n_global=n_global, want_loss=want_loss)
            sl = self._slice_of(self._bucket_ranges[k][1]) if self.shard_optimizer else None
            if sl is not None:
                works.append((self._reduce_scatter(self._buckets[k], sl[0], sl[1]), sl))
            else:
                works.append((self.dist.all_reduce(self._buckets[k], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), None))
        if not self.shard_optimizer:
            for w, _ in works:
                w.wait()
            if want_loss and loss is not None:
                loss = self._reduce_loss(loss)
            self.backend.apply_gradients()
            return loss
        # sharded optimizer step: as each bucket's sum arrives, Adam on the owned slice, then the slice travels to the others
        gathers, first = [], True
        for k, (w, sl) in enumerate(works):
            w.wait()
            off, n = self._bucket_ranges[k]
            if sl is None:                               # bucket not cut: every rank updates all of it
                self.backend.apply_gradients_range(off, n, first)
            else:
                self.backend.apply_gradients_range(off + sl[0], sl[1], first)
                pb = self._param[off:off + n]
                mine = pb[sl[0]:sl[0] + sl[1]]           # RCCL: the in-place form (input = this rank's slice of the output)
                if self.dist.get_backend(self.group) != "nccl":
                    mine = mine.clone()
                gathers.append(self.dist.all_gather_into_tensor(pb, mine, group=self.group, async_op=True))
            first = False
        if want_loss and loss is not None:
            loss = self._reduce_loss(loss)
        for g in gathers:
            g.wait()
        if hasattr(self.backend, "params_changed"):
            self.backend.params_changed()
        return loss

    def gather_optimizer_state(self):
        """shard_optimizer: Adam's moments are only kept up to date on the slice their rank owns.  -> (m, v, iterations) with
        every slice taken from its owner (numpy, on every rank): what a checkpoint of the job has to hold."""
        m, v, it = self.backend.get_optimizer_state()
        if not self.shard_optimizer or self.world == 1:
            return m, v, it
        import torch
        dev = self._grad.device if self._grad is not None else "cpu"
        out = []
        for arr in (m, v):
            t = torch.as_tensor(np.ascontiguousarray(arr), device=dev)
            for off, n in (self._bucket_ranges or self.backend.grad_buckets()):
                sl = self._slice_of(n)
                if sl is None:
                    continue
                b = t[off:off + n]
                self.dist.all_gather_into_tensor(b, b[sl[0]:sl[0] + sl[1]].clone(), group=self.group)
            out.append(t.cpu().numpy())
        return out[0], out[1], it

    def all_reduce_numpy(self, arr):
        """Sum of a small float64 numpy array over the ranks (statistics, not the hot path).  RCCL has no CPU tensors:
        the array travels through the device the gradient lives on; gloo takes the CPU tensor as is."""
        if self.world == 1:
            return np.asarray(arr, np.float64)
        import torch
        if self._grad is None:
            self._grad = self.backend.grad_tensor()
        t = torch.as_tensor(np.asarray(arr, np.float64), device=self._grad.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()
