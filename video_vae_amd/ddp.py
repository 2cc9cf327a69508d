"""Data-parallel gradient reduction: bucketed all-reduce over RCCL/xGMI overlapped with backward.

Replaces the implicit XLA SPMD gradient all-reduce of the reference's TPU path
(claude_distributed/distributed_train.py:107-109,378-382,412: params P() replicated, batch P('data')).
One process per GPU; ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests) is
only the transport -- bucketing and overlap are owned here and in optim.Optimizer:

* gradients land in one flat fp32 buffer (Optimizer.g) laid out in reverse registration order, so the buckets are
  contiguous slices that fill front-to-back as backward proceeds (UNet first, encoder last);
* the optimizer's per-parameter hook counts arrivals; when a bucket is complete it is copied into the flat buffer and
  ``launch(b)`` all-reduces that slice (SUM, async) at once -- RCCL runs on its own stream while the backward conv
  stack and the remaining transformer layers continue;
* ``finish()`` (called by Optimizer.update) waits for the outstanding collectives; the 1/world mean and the
  global-norm clip are folded into the fused Adam kernel, so no extra pass touches the gradients.

Replicas stay bit-identical: every rank applies the same update to the same all-reduced buffer
(the property the reference checks at claude_distributed/test_distributed.py:159-163).

``grad_dtype=torch.bfloat16`` (train.py / bench.py ``--grad-dtype bf16``) halves what crosses xGMI: a landed bucket is rounded to bf16
(one cast launch), all-reduced in bf16 and widened back into the fp32 buffer behind the wait.  A ring all-reduce moves
2 (N - 1) / N x 683 MB per GPU in fp32 over point-to-point links of ~153 GB/s each; at N = 2 and 4, where one or three links carry it,
that is the part of the step the backward cannot hide.  The price is the gradient's mantissa (8 bits per addend, the sum kept in
bf16 by the transport); replicas still hold identical values.  Default fp32 = the reference's arithmetic.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, optimizer, process_group=None, grad_dtype=torch.float32):
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype is torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        self.opt = optimizer
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.handles = []
        # RCCL orders a collective after the work already queued on the current stream and its result before what is queued
        # next; gloo's device path (CPU-side tests of this class with GPU tensors) is fenced by hand instead
        self.fence = optimizer.p.is_cuda and dist.get_backend(process_group) != "nccl"
        optimizer.reducer = self

    def launch(self, b):
        s, e = self.opt.buckets[b]
        if self.fence:
            torch.cuda.synchronize()
        if self.grad_dtype == torch.float32:
            self.handles.append((dist.all_reduce(self.opt.g[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, s, e))
        else:
            low = self.opt.g[s:e].to(self.grad_dtype)                  # rounded once on this rank; the transport sums in bf16
            if self.fence:
                torch.cuda.synchronize()
            self.handles.append((dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self.group, async_op=True), low, s, e))

    def reset(self):
        self.handles = []

    def finish(self):
        for h, low, s, e in self.handles:
            h.wait()
            if low is not None:
                if self.fence:
                    torch.cuda.synchronize()
                self.opt.g[s:e].copy_(low)                             # widened back behind the wait, on the current stream
        if self.fence and self.handles:
            torch.cuda.synchronize()
        self.handles = []

    def broadcast_state(self, src=0):
        """Replicate rank ``src``'s parameters AND optimizer state -- Adam moments and the update count that indexes the
        learning-rate schedule and the bias correction -- the reference's resume broadcast of {"model", "optimizer"}
        (claude_distributed/distributed_train.py:321-341) and its device_put(state, P()) at start-up (:378-380).  Without the
        moments and the count, ranks other than ``src`` would apply a different update from the first step after a resume."""
        opt = self.opt
        cuda = opt.p.is_cuda
        if cuda:
            torch.cuda.synchronize()          # one-time setup: no reliance on the transport's ordering against in-flight work
        count = torch.tensor([opt.count], dtype=torch.int64, device=opt.p.device)
        for t in (opt.p, opt.m, opt.v, count):
            dist.broadcast(t, src=src, group=self.group)
        if cuda:
            torch.cuda.synchronize()
        opt.count = int(count.item())
        opt.refresh_shadow()

    def broadcast_parameters(self, src=0):
        """Kept name: replicates the whole training state (see broadcast_state)."""
        self.broadcast_state(src)


def all_reduce_mean_scalars(values, process_group=None):
    """Global mean of per-rank scalar metrics (equal shard sizes => mean of means, distributed_train.py:474)."""
    t = torch.stack([v.detach().float().reshape(()) for v in values])
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    return list((t / dist.get_world_size(process_group)).unbind())
