"""CPU restatement of the reference's video discriminator (train/classifier.py).  Test infrastructure only.

Classifier = ConvBlock3D(c -> F, 3x7x7) + num_levels x DownBlock3D(F 2^i -> F 2^(i+1)) (no skips) + mean over (t, h, w) + Linear(-> 1),
every convolution wrapped in ManualSpectralNorm: one power-iteration step on the (kt kh kw Cin, Cout) matrix view of the kernel,
kernel / sigma for this call only, the iterate ``u`` kept as state (classifier.py:10-67).  The iteration is computed from the kernel
inside the differentiated function, so gradients flow through sigma, u and v as they do under nnx.value_and_grad.
Parameters: flat ``{dotted.name: tensor}`` as in oracle/unet.py, state ``{"<block>.conv.u": (1, Cout)}``.
"""
import torch

from . import nn as O
from .unet import sub


def spectral_norm_kernel(kernel, u, n_steps=1, update_stats=True):
    """-> (kernel / sigma, new u).  classifier.py:22-56."""
    w = kernel.reshape(-1, kernel.shape[-1])
    v = None
    if update_stats:
        for _ in range(n_steps):
            v = u @ w.T
            v = v / torch.linalg.norm(v)
            u = v @ w
            u = u / torch.linalg.norm(u)
    if v is None:
        v = u @ w.T
        v = v / torch.linalg.norm(v)
    sigma = ((v @ w) @ u.T)[0, 0]
    return kernel / sigma, u


def conv_block3d_sn(p, state, prefix, x, dtype=O.F32, update_stats=True):
    """ConvBlock3D of classifier.py:69-92 -> (y, new u)."""
    k, u = spectral_norm_kernel(p[f"{prefix}.conv.kernel"], state[f"{prefix}.conv.u"], 1, update_stats)
    cout = k.shape[-1]
    x = O.conv3d_same(x, k, p[f"{prefix}.conv.bias"], dtype)
    x = O.group_norm(x, p[f"{prefix}.norm.scale"], p[f"{prefix}.norm.bias"], min(8, cout), dtype)
    return O.silu(x, dtype), u


def classifier_num_levels(p):
    n = 0
    while f"encoders.{n}.conv1.conv.kernel" in p:
        n += 1
    return n


def classifier(p, state, x, dtype=O.F32, update_stats=True):
    """Classifier.__call__ (classifier.py:155-179) -> (logits (b, 1), new state)."""
    new = {}
    x = O.q(x, dtype)
    x, new["initial_conv.conv.u"] = conv_block3d_sn(p, state, "initial_conv", x, dtype, update_stats)
    for i in range(classifier_num_levels(p)):
        for c in ("conv1", "conv2"):
            name = f"encoders.{i}.{c}"
            x, new[f"{name}.conv.u"] = conv_block3d_sn(p, state, name, x, dtype, update_stats)
        x = O.max_pool_1x2x2(x)
    x = O.q(x.mean(dim=(1, 2, 3)), dtype)
    return O.linear(x, p["classifier.kernel"], p["classifier.bias"], dtype), {k: v.detach() for k, v in new.items()}


def init_classifier(channels, base_features=32, num_levels=4, seed=0, temporal_kernel=3):
    """Parameter tree and spectral-norm state of Classifier.__init__ (classifier.py:122-153)."""
    from .unet import init_conv_block
    gen = torch.Generator().manual_seed(seed)
    p, state = {}, {}
    init_conv_block(p, "initial_conv", channels, base_features, 7, temporal_kernel, gen)
    cin = base_features
    for i in range(num_levels):
        cout = base_features * 2 ** (i + 1)
        init_conv_block(p, f"encoders.{i}.conv1", cin, cout, 3, temporal_kernel, gen)
        init_conv_block(p, f"encoders.{i}.conv2", cout, cout, 3, temporal_kernel, gen)
        cin = cout
    p["classifier.kernel"] = O.lecun_normal_((cin, 1), cin, gen)
    p["classifier.bias"] = torch.zeros(1)
    for k in list(p):
        if k.endswith("conv.kernel"):
            state[k[:-6] + "u"] = torch.randn((1, p[k].shape[-1]), generator=gen)
    return p, state
