"""Stand-in for flax.nnx.Rngs: counter-based keys per stream + a hook to inject explicit noise.

The reference threads an ``nnx.Rngs`` through every stochastic op and calls
``rngs.sampling()`` / ``rngs.params()`` once per draw (train/model.py:106,125;
train/layers.py:244).  JAX threefry streams cannot be reproduced in torch, so
only the distributions are matched; for parity tests the caller injects the
noise tensors by name (``rngs.inject("reparam_eps", eps)``).
"""
import hashlib

import torch


class Key:
    def __init__(self, seed, stream, count):
        self.seed, self.stream, self.count = seed, stream, count

    def generator(self, device="cpu"):
        h = hashlib.sha256(f"{self.seed}/{self.stream}/{self.count}".encode()).digest()
        g = torch.Generator(device=device)
        g.manual_seed(int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF)
        return g


class Rngs:
    """``Rngs(seed)``; ``rngs.sampling()`` / ``rngs.params()`` return a fresh Key each call."""

    def __init__(self, seed=0):
        self.seed = int(seed)
        self.counts = {}
        self.overrides = {}
        self.recording = None          # dict: name -> (kind, shape, dtype) of every draw (used by graph.GraphedTrainStep)

    def _key(self, stream):
        c = self.counts.get(stream, 0)
        self.counts[stream] = c + 1
        return Key(self.seed, stream, c)

    def sampling(self):
        return self._key("sampling")

    def params(self):
        return self._key("params")

    def __call__(self):
        return self._key("default")

    # ---- explicit-noise hook (parity tests) ----
    def inject(self, name, tensor):
        self.overrides[name] = tensor

    def clear(self):
        self.overrides.clear()

    def draw(self, name, kind, shape, device, dtype=torch.float32):
        """Noise for the op called ``name``: the injected tensor if present, else a fresh draw.

        A key is consumed either way so the stream position matches the reference's call order.
        """
        key = self.sampling()
        if self.recording is not None:
            self.recording[name] = (kind, tuple(shape), dtype)
        if name in self.overrides:
            t = self.overrides[name]
            assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
            return t.to(device=device, dtype=dtype)
        g = key.generator(device)
        if kind == "normal":
            return torch.randn(shape, generator=g, device=device, dtype=dtype)
        if kind == "uniform":
            return torch.rand(shape, generator=g, device=device, dtype=dtype)
        raise ValueError(kind)


def truncated_normal_(shape, fan_in, key, scale=1.0):
    """variance_scaling(scale, 'fan_in', 'truncated_normal') = Flax lecun_normal when scale=1."""
    std = (scale / fan_in) ** 0.5 / 0.87962566103423978
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=key.generator("cpu"))
    return t * std
