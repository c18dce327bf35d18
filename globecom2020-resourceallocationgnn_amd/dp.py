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

    def __init__(self, backend, process_group=None, force=False, overlap=None):
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
        self.overlap = (os.environ.get("V2X_DP_OVERLAP", "0") == "1") if overlap is None else bool(overlap)

    def shard(self, batch, y):
        """Contiguous shard of whole graphs for this rank (variable-size batches: balanced by edges + nodes,
        PackedBatch.shard_bounds).  y is [R, C] for the full batch."""
        if self.world == 1:
            return batch, y
        sh, (r0, r1) = batch.shard(self.rank, self.world, with_rows=True)
        return sh, y[r0:r1]

    def _overlapped(self, local_batch):
        """Bucketed all-reduce overlapped with the backward pass: available when the backend splits its step
        (GnnEngine.forward_backward_phase) and the batch is resident on the device."""
        return (self.overlap and hasattr(self.backend, "forward_backward_phase") and hasattr(local_batch, "device")
                and getattr(self.backend.spec, "feat_dim", 0) <= 64)

    def train_step(self, local_batch, local_y, n_graphs_global=None, want_loss=True, n_denominator=None):
        """forward+backward on the local shard, all-reduce the gradient, Adam on every rank.

        n_denominator (n_graphs_global is its older name): what the GLOBAL Huber mean divides by besides the channel
        count -- the number of graphs of the whole minibatch for fixed-size graphs (one mean per output, BS_brain.py:214),
        the number of NODE ROWS of the whole minibatch for variable_graphs models (one mean over all rows): ragged shards
        hold different numbers of rows, so the caller sums them over the ranks (bench.py does).

        With `overlap` and a device-resident batch the step is split (SURVEY.md 8 e3 "overlappable with the tail of
        backward"): the Dense-layer gradients are final as soon as the decision MLP has been differentiated, so their
        bucket (the tail of the flat gradient) is all-reduced -- asynchronously, RCCL works on its own stream -- while
        the graph layers are still in their backward pass; the graph-layer bucket follows, and ONE Adam launch runs
        after both.  Otherwise: one replayed graph for forward + backward + slab sums, ONE all-reduce of the flat
        gradient, Adam."""
        if n_denominator is not None:
            n_graphs_global = n_denominator
        if n_graphs_global is None:
            raise ValueError("train_step: the global Huber denominator (n_denominator) is required under data parallelism")
        reduce_now = self.world > 1 or self.force
        if reduce_now and self._overlapped(local_batch):
            if self._grad is None:
                self._grad = self.backend.grad_tensor()
                self._buckets = [self._grad[o:o + n] for o, n in self.backend.grad_buckets()]
            self.backend.forward_backward_phase(local_batch, local_y, 0, n_global=n_graphs_global)
            w0 = self.dist.all_reduce(self._buckets[0], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            loss = self.backend.forward_backward_phase(local_batch, local_y, 1, n_global=n_graphs_global, want_loss=want_loss)
            w1 = self.dist.all_reduce(self._buckets[1], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            w0.wait()
            w1.wait()
            if want_loss and loss is not None:
                self.dist.all_reduce(loss, op=self.dist.ReduceOp.SUM, group=self.group)
            self.backend.apply_gradients()
            return loss
        loss = self.backend.forward_backward(local_batch, local_y, n_global=n_graphs_global, want_loss=want_loss)
        if reduce_now:
            if self._grad is None:
                self._grad = self.backend.grad_tensor()
            self.dist.all_reduce(self._grad, op=self.dist.ReduceOp.SUM, group=self.group)
            if want_loss and loss is not None:
                import torch
                if not torch.is_tensor(loss):
                    # host-side loss (numpy batches): reduce it on the device the gradient lives on (RCCL has no
                    # CPU tensors; gloo takes the CPU tensor as is)
                    loss_t = torch.as_tensor(np.asarray(loss, np.float64), device=self._grad.device)
                    self.dist.all_reduce(loss_t, op=self.dist.ReduceOp.SUM, group=self.group)
                    loss = loss_t.cpu().numpy()
                else:
                    self.dist.all_reduce(loss, op=self.dist.ReduceOp.SUM, group=self.group)
        self.backend.apply_gradients()
        return loss

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
