"""optax.chain(clip_by_global_norm(1.0), adam(warmup_cosine_decay_schedule)) on flat fp32 buffers.

Reference: train/rl_nonadversarial.py:241-253.  All parameters live in ONE flat fp32 buffer (params are views into
it), and so do the gradients and Adam moments: the whole update is two HIP launches (squared-norm reduction, fused
clip+Adam, which also refreshes a bf16 shadow copy of the weights for the next forward) regardless of the parameter
count.  The buffer is laid out in REVERSE registration order and cut into buckets, so the gradients that become ready
first in backward (the UNet at the decoder tail) fill the first bucket.

Gradients land in the flat buffer bucket by bucket: a post-accumulate hook per parameter counts arrivals; when a
bucket is complete its gradients are copied (and widened to fp32) into their slots with one fused multi-tensor copy
and the autograd-owned tensors are released -- no zero-fill of the buffer, no per-parameter accumulate kernels --
and, under data parallelism, the bucket's slice is all-reduced right away (ddp.GradReducer).
"""
import ctypes
import math

import torch

from ._lib import lib, check


def warmup_cosine_decay_schedule(init_value, peak_value, warmup_steps, decay_steps, end_value):
    """optax.warmup_cosine_decay_schedule: returns schedule(count), count 0-based."""
    def schedule(count):
        if count < warmup_steps:
            return init_value + (peak_value - init_value) * (count / warmup_steps)
        span = decay_steps - warmup_steps
        c = min(count - warmup_steps, span)
        cosine = 0.5 * (1.0 + math.cos(math.pi * c / span))
        alpha = end_value / peak_value
        return peak_value * ((1 - alpha) * cosine + alpha)
    return schedule


def reference_schedule(batch_size=2, learning_rate=2e-5, decay_steps=1_000_000):
    """The reference's schedule constants (rl_nonadversarial.py:44-52,241-247)."""
    return warmup_cosine_decay_schedule(0.0, learning_rate, 20000 // math.sqrt(batch_size), decay_steps, learning_rate / 10)


def _copy_all(dsts, srcs):
    """Gradients -> their slots of the flat buffer: one grouped HIP launch per 64 tensors (a memcpy node each costs ~4 us of
    GPU time inside the replayed graph), the framework's foreach copy for anything that is not a plain fp32 GPU range."""
    from . import ops
    fast = [(d, s) for d, s in zip(dsts, srcs) if ops.copy_grouped_ok(d, s)]
    rest = [(d, s) for d, s in zip(dsts, srcs) if not ops.copy_grouped_ok(d, s)]
    if fast:
        ops.copy_grouped([d for d, _ in fast], [s for _, s in fast])
    if rest:
        torch._foreach_copy_([d for d, _ in rest], [s for _, s in rest])


STREAM_TAG = "vvae_accumulate_stream"


def accumulate_grad_node(param):
    """The live AccumulateGrad node of a leaf parameter (made on the current stream if none is alive), or None."""
    with torch.enable_grad():
        fn = param.expand_as(param).grad_fn
    nxt = fn.next_functions if fn is not None else ()
    return nxt[0][0] if nxt and nxt[0][0] is not None else None


class Optimizer:
    """Counterpart of ``nnx.Optimizer(model, optax.chain(clip_by_global_norm(max_norm), adam(schedule)))``.

    ``zero_grad()`` then backward then ``update()``.  After ``update()`` the gradients of the step are in ``self.g``.
    """

    def __init__(self, model, schedule, max_norm=1.0, b1=0.9, b2=0.999, eps=1e-8, bf16_shadow=True,
                 bucket_bytes=64 << 20):
        self.model = model
        self.schedule = schedule if callable(schedule) else (lambda count, lr=schedule: lr)
        self.max_norm, self.b1, self.b2, self.eps = max_norm, b1, b2, eps
        self.count = 0
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in named][::-1]
        self.params = [p for _, p in named][::-1]
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32:
                raise ValueError("Optimizer expects fp32 parameters (param_dtype=float32)")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # 16-byte aligned slots
        self.numel = off
        self.p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.gnorm_part = torch.zeros(max(1, lib().vvae_sqnorm_blocks(off)) if dev.type == "cuda" else 1, dtype=torch.float64, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if bf16_shadow else None
        self.gviews = []
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + n].view(p.shape)
            p.grad = None
            self.gviews.append(self.g[o:o + n].view(p.shape))
            p.gview = self.gviews[-1]                             # written directly by ops.deferred_wgrad
            if self.shadow is not None:
                p.bf16 = self.shadow[o:o + n].view(p.shape)       # read by layers.Linear: no per-call weight casts
        if self.shadow is not None:
            self.shadow.copy_(self.p)
        # transposed bf16 shadows (out, in) of the Linear kernels whose forward runs on the own NT GEMM (layers mark them ``want_t``:
        # the MLP's fc1, whose product also emits SiLU); refreshed after every update from the bf16 shadow, one grouped launch per 64
        self.tpairs = []
        if self.shadow is not None and dev.type == "cuda":
            # (the grouped transpose wants 16-byte aligned operands: a shadow slot behind an odd-sized parameter is 8-byte aligned only -- that layer
            #  then simply has no transposed shadow and its forward runs the library product)
            want = [p for p in self.params if getattr(p, "want_t", False) and p.dim() == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0
                    and p.bf16.data_ptr() % 16 == 0]
            if want:
                tbuf = torch.zeros(sum(p.numel() for p in want), dtype=torch.bfloat16, device=dev)
                o = 0
                for p in want:
                    p.bf16_t = tbuf[o:o + p.numel()].view(p.shape[1], p.shape[0])
                    self.tpairs.append((p.bf16, p.bf16_t))
                    o += p.numel()
                self._refresh_transposed()
        # ---- buckets: contiguous [start, end) element ranges aligned to parameter slots ----
        cap = max(1, bucket_bytes // 4)
        ends = [o + (p.numel() + 3) // 4 * 4 for o, p in zip(self.offsets, self.params)]
        self.buckets, self.param_bucket, self.bucket_params = [], [], []
        start, members = 0, []
        for i, e in enumerate(ends):
            self.param_bucket.append(len(self.buckets))
            members.append(i)
            if e - start >= cap or i == len(ends) - 1:
                self.buckets.append((start, e))
                self.bucket_params.append(members)
                start, members = e, []
        self.arrived = [0] * len(self.buckets)
        self.landed = [False] * len(self.buckets)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.external = set()          # parameter indices whose flat-buffer slot was written directly this step
        self.clean = set(range(len(self.params)))     # slots known to hold zeros (self.g starts zeroed)
        self.hooks_active = True       # False while gradients arrive through land_all (graph capture) instead of the hooks
        self.reducer = None            # set by ddp.GradReducer
        self.defer_reduce = False      # graph mode: do not launch collectives from the landing hooks (see graph.py)
        self.prelaunched = set()       # graph mode: buckets already handed to the reducer between the two replays of a step
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    # ---- gradient landing -------------------------------------------------------------------------------
    def _make_hook(self, i):
        tagged = [False]

        def hook(param):
            if not tagged[0] and param.is_cuda:
                # Remember, ON the parameter's AccumulateGrad node, which stream it is pinned to (the engine runs the node -- and this
                # hook -- on the stream that was current when the node was made).  graph.GraphedTrainStep reads the tag back: a capture
                # on another stream while such a node is alive crashes the autograd engine (the r03b incident).
                tagged[0] = True
                node = accumulate_grad_node(param)
                if node is not None:
                    node.metadata.setdefault(STREAM_TAG, torch.cuda.current_stream(param.device).cuda_stream)
            if param.grad is None:               # the engine also runs the hook for an undefined gradient (a backward that returned
                return                           # None: parked weight gradients) -- those arrive through mark_external instead
            b = self.param_bucket[i]
            self.arrived[b] += 1
            if self.arrived[b] == len(self.bucket_params[b]) and not self.landed[b]:
                self._land(b)
        return hook

    @torch.no_grad()
    def _land(self, b):
        """Move bucket b's gradients into the flat buffer (missing ones = zeros) and hand the slice to the reducer."""
        dsts, srcs = [], []
        for i in self.bucket_params[b]:
            p = self.params[i]
            if i in self.external:
                self.clean.discard(i)                             # already in place (ops.deferred_wgrad)
            elif p.grad is None:
                if i not in self.clean:                           # a slot nobody has written since it was last zeroed stays zero
                    self.gviews[i].zero_()
                    self.clean.add(i)
            elif p.grad.data_ptr() != self.gviews[i].data_ptr():
                self.clean.discard(i)
                dsts.append(self.gviews[i])
                srcs.append(p.grad)
            else:
                self.clean.discard(i)
            p.grad = None
        if dsts:
            _copy_all(dsts, srcs)
        self.landed[b] = True
        if self.reducer is not None and not self.defer_reduce:
            self.reducer.launch(b)

    @torch.no_grad()
    def land_all(self, grads):
        """Install gradients given in ``self.params`` order (None = zero), e.g. from torch.autograd.grad (graph capture)."""
        dsts, srcs = [], []
        for i, (gv, gr) in enumerate(zip(self.gviews, grads)):
            if i in self.external:
                self.clean.discard(i)
                continue                                          # already in place (ops.deferred_wgrad)
            if gr is None:
                if i not in self.clean:                           # a slot nobody has written since it was last zeroed stays zero
                    gv.zero_()
                    self.clean.add(i)
            else:
                self.clean.discard(i)
                dsts.append(gv)
                srcs.append(gr)
        if dsts:
            _copy_all(dsts, srcs)
        self.landed = [True] * len(self.buckets)
        if self.reducer is not None and not self.defer_reduce:      # same contract as _land: a landed bucket goes to the reducer
            for b in range(len(self.buckets)):
                self.reducer.launch(b)

    @torch.no_grad()
    def land_subset(self, indices, grads):
        """land_all for a subset of ``self.params`` (indices, gradients in the same order); buckets are not marked landed."""
        dsts, srcs = [], []
        for i, gr in zip(indices, grads):
            if i in self.external:
                self.clean.discard(i)
                continue
            if gr is None:
                if i not in self.clean:
                    self.gviews[i].zero_()
                    self.clean.add(i)
            else:
                self.clean.discard(i)
                dsts.append(self.gviews[i])
                srcs.append(gr)
        if dsts:
            _copy_all(dsts, srcs)

    def mark_external(self, param):
        """The gradient of ``param`` has been written straight into its flat-buffer slot for this step (ops.deferred_wgrad):
        it counts as arrived, so a completed bucket goes to the reducer while backward is still running."""
        i = self.index[id(param)]
        self.external.add(i)
        if self.hooks_active:
            b = self.param_bucket[i]
            self.arrived[b] += 1
            if self.arrived[b] == len(self.bucket_params[b]) and not self.landed[b]:
                self._land(b)

    def zero_grad(self):
        self.external = set()
        for p in self.params:
            p.grad = None
        self.arrived = [0] * len(self.buckets)
        self.landed = [False] * len(self.buckets)
        if self.reducer is not None:
            self.reducer.reset()

    def set_grads(self, grads):
        """Test helper: install explicit gradients {name: tensor} as if backward had produced them."""
        self.zero_grad()
        by_name = dict(zip(self.names, range(len(self.names))))
        for n, gr in grads.items():
            i = by_name[n]
            self.params[i].grad = gr.to(self.g.device)
        for b in range(len(self.buckets)):
            self._land(b)

    @torch.no_grad()
    def update(self):
        """optimizer.update(grads): clip by global norm, then Adam with lr = schedule(count)."""
        for b in range(len(self.buckets)):
            if not self.landed[b]:
                self._land(b)                    # parameters that received no gradient this step contribute zeros
        gscale = 1.0
        if self.reducer is not None:
            if self.defer_reduce:
                if not self.prelaunched:
                    self.reducer.reset()
                for b in range(len(self.buckets)):
                    if b not in self.prelaunched:
                        self.reducer.launch(b)
                self.prelaunched = set()
            self.reducer.finish()
            gscale = 1.0 / self.reducer.world_size
        if not self.p.is_cuda:
            raise RuntimeError("Optimizer.update runs the fused HIP clip+Adam kernel and needs GPU parameters")
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lr = float(self.schedule(self.count))
        self.count += 1
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        # global norm without atomics: per-workgroup partial sums of squares, folded in one fixed order inside the Adam kernel
        check(lib().vvae_sqnorm_partials(vp(self.g), self.numel, vp(self.gnorm_part), s), "vvae_sqnorm_partials")
        check(lib().vvae_adam_clip_step(vp(self.p), vp(self.g), vp(self.m), vp(self.v),
                                        vp(self.shadow) if self.shadow is not None else None, self.numel, vp(self.gnorm_part),
                                        self.gnorm_part.numel(), vp(self.gnorm_sq), gscale, self.max_norm, lr, self.b1, self.b2,
                                        self.eps, self.count, s),
              "vvae_adam_clip_step")
        if self.tpairs:
            self._refresh_transposed()
        self.last_lr = lr
        return lr

    def _refresh_transposed(self):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i0 in range(0, len(self.tpairs), 64):
            e = self.tpairs[i0:i0 + 64]
            n = len(e)
            VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
            check(lib().vvae_transpose_grouped_bf16(VP(*[a.data_ptr() for a, _ in e]), VP(*[b.data_ptr() for _, b in e]),
                                                    IA(*[a.shape[0] for a, _ in e]), IA(*[a.shape[1] for a, _ in e]), n, s),
                  "vvae_transpose_grouped_bf16")

    def refresh_shadow(self):
        """Re-derive the bf16 shadows after parameters were written from outside (checkpoint load, broadcast)."""
        if self.shadow is not None:
            self.shadow.copy_(self.p)
            if self.tpairs:
                self._refresh_transposed()

    def grad_norm(self):
        """||g|| of the last update (host sync)."""
        scale = 1.0 / self.reducer.world_size if self.reducer is not None else 1.0
        return float(self.gnorm_sq.item()) ** 0.5 * scale

    # ---- checkpoint state (model_loader.save_checkpoint / load_checkpoint) ----
    def state_dict(self):
        out = {"count": self.count}
        for n, p, o in zip(self.names, self.params, self.offsets):
            k = p.numel()
            out[f"mu.{n}"] = self.m[o:o + k].view(p.shape).detach().cpu().clone()
            out[f"nu.{n}"] = self.v[o:o + k].view(p.shape).detach().cpu().clone()
        return out

    def load_state_dict(self, state):
        self.count = int(state["count"])
        for n, p, o in zip(self.names, self.params, self.offsets):
            k = p.numel()
            self.m[o:o + k].copy_(state[f"mu.{n}"].reshape(-1))
            self.v[o:o + k].copy_(state[f"nu.{n}"].reshape(-1))
        self.refresh_shadow()
