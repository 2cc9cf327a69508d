"""optax.chain(clip_by_global_norm(1.0), adam(warmup_cosine_decay_schedule)) on flat fp32 buffers.

Reference: train/rl_nonadversarial.py:241-253.  All parameters live in ONE flat fp32 buffer (params are views into
it), and so do gradients, Adam moments: the whole update is two HIP launches (squared-norm reduction, fused
clip+Adam) regardless of the parameter count, and a data-parallel reducer (ddp.py) all-reduces slices of the
flat gradient buffer in place.  The buffer is laid out in REVERSE registration order so that the gradients that
become ready first in backward (the UNet at the decoder tail) sit at the front of the first bucket.
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


class Optimizer:
    """Counterpart of ``nnx.Optimizer(model, optax.chain(clip_by_global_norm(max_norm), adam(schedule)))``.

    ``update()`` applies one step from the gradients accumulated in ``.grad`` (which alias ``self.g``).
    """

    def __init__(self, model, schedule, max_norm=1.0, b1=0.9, b2=0.999, eps=1e-8, bf16_shadow=False):
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
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if bf16_shadow else None
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + n].view(p.shape)
            p.grad = self.g[o:o + n].view(p.shape)
        self.reducer = None            # set by ddp.GradReducer
        self.last_grad_norm = None

    def zero_grad(self):
        self.g.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * o:
                p.grad = self.g[o:o + p.numel()].view(p.shape)
        if self.reducer is not None:
            self.reducer.reset()

    @torch.no_grad()
    def update(self):
        """optimizer.update(grads): clip by global norm, then Adam with lr = schedule(count)."""
        gscale = 1.0
        if self.reducer is not None:
            self.reducer.finish()
            gscale = 1.0 / self.reducer.world_size
        if not self.p.is_cuda:
            raise RuntimeError("Optimizer.update runs the fused HIP clip+Adam kernel and needs GPU parameters")
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lr = float(self.schedule(self.count))
        self.count += 1
        self.gnorm_sq.zero_()
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        check(lib().vvae_sqnorm_accum(vp(self.g), self.numel, vp(self.gnorm_sq), s), "vvae_sqnorm_accum")
        check(lib().vvae_adam_clip_step(vp(self.p), vp(self.g), vp(self.m), vp(self.v),
                                        vp(self.shadow) if self.shadow is not None else None, self.numel, vp(self.gnorm_sq),
                                        gscale, self.max_norm, lr, self.b1, self.b2, self.eps, self.count, s),
              "vvae_adam_clip_step")
        self.last_lr = lr
        return lr

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
