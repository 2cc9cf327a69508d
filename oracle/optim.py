"""CPU restatement of the optax optimiser chain.  Test infrastructure only.

train/rl_nonadversarial.py:241-253: chain(clip_by_global_norm(1.0),
adam(warmup_cosine_decay_schedule(0 -> 2e-5 over 14142 steps -> 2e-6 at 1e6))).
Semantics per SURVEY.md A.13.
"""
import math

import torch


def warmup_cosine_decay_schedule(count, init_value, peak_value, warmup_steps, decay_steps, end_value):
    """optax.warmup_cosine_decay_schedule evaluated at update ``count`` (0-based)."""
    if count < warmup_steps:
        return init_value + (peak_value - init_value) * (count / warmup_steps)
    c = min(count - warmup_steps, decay_steps - warmup_steps)
    cosine = 0.5 * (1.0 + math.cos(math.pi * c / (decay_steps - warmup_steps)))
    alpha = end_value / peak_value
    return peak_value * ((1 - alpha) * cosine + alpha)


def clip_by_global_norm(grads, max_norm):
    """optax.clip_by_global_norm: scale by max_norm/||g|| only when ||g|| >= max_norm (no epsilon)."""
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    if gn < max_norm:
        return dict(grads), gn
    return {k: g / gn * max_norm for k, g in grads.items()}, gn


class Adam:
    """optax.adam(b1=.9, b2=.999, eps=1e-8, eps_root=0) with bias correction; state per param."""

    def __init__(self, params, b1=0.9, b2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps = b1, b2, eps
        self.count = 0
        self.mu = {k: torch.zeros_like(v) for k, v in params.items()}
        self.nu = {k: torch.zeros_like(v) for k, v in params.items()}

    def update(self, params, grads, lr):
        self.count += 1
        c1 = 1 - self.b1 ** self.count
        c2 = 1 - self.b2 ** self.count
        out = {}
        for k, g in grads.items():
            self.mu[k] = self.b1 * self.mu[k] + (1 - self.b1) * g
            self.nu[k] = self.b2 * self.nu[k] + (1 - self.b2) * g * g
            upd = (self.mu[k] / c1) / (torch.sqrt(self.nu[k] / c2) + self.eps)
            out[k] = params[k] - lr * upd
        return out


def train_update(params, grads, adam, schedule_kwargs, max_norm=1.0):
    """One optimizer.update(grads): clip -> adam with lr = schedule(count before increment)."""
    lr = warmup_cosine_decay_schedule(adam.count, **schedule_kwargs)
    clipped, gn = clip_by_global_norm(grads, max_norm)
    return adam.update(params, clipped, lr), gn, lr
