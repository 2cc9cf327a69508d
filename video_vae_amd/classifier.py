"""Video real/fake discriminator with the reference's class surface (train/classifier.py), on the HIP kernels.

``Classifier(channels, base_features=32, num_levels=4, rngs, temporal_kernel=3, dtype, param_dtype)(x, mask=None, train=True)``
-> (b, 1) logits: a spectrally normalised ConvBlock3D stack (3x7x7 stem, then ``num_levels`` x [two 3x3x3 blocks + spatial
max-pool]), the mean over (t, h, w) and a Linear head.  Convolutions, GroupNorm+SiLU and pooling are the library's kernels (ops);
the power iteration works on (kt kh kw Cin, Cout) matrices of a few thousand elements and stays in torch fp32.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .rngs import Rngs, truncated_normal_
from .unet import Conv, GroupNorm


class ManualSpectralNorm(nn.Module):
    """classifier.py:10-67: one power-iteration step per call on the wrapped layer's kernel; ``u`` is state (a buffer), the
    kernel is divided by sigma for this call only, and autograd sees the whole computation (sigma, u and v depend on the kernel)."""

    def __init__(self, layer, rngs, n_steps=1):
        super().__init__()
        self.layer = layer
        self.n_steps = n_steps
        self.register_buffer("u", torch.randn((1, layer.kernel.shape[-1]), generator=rngs.params().generator("cpu")))

    def normalised_kernel(self, update_stats=True):
        weight = self.layer.kernel.float()
        w = weight.reshape(-1, weight.shape[-1])
        u, v = self.u, None
        if update_stats:
            for _ in range(self.n_steps):
                v = u @ w.T
                v = v / torch.linalg.norm(v)
                u = v @ w
                u = u / torch.linalg.norm(u)
            self.u = u.detach()
        if v is None:
            v = u @ w.T
            v = v / torch.linalg.norm(v)
        sigma = ((v @ w) @ u.T)[0, 0]
        return weight / sigma

    def forward(self, x, update_stats=True):
        return ops.conv3d(x.to(self.layer.dtype), self.normalised_kernel(update_stats), self.layer.bias)


class ConvBlock3D(nn.Module):
    """Spectrally normalised Conv(kt, k, k) SAME -> GroupNorm(min(8, C)) -> SiLU.  classifier.py:69-92."""

    def __init__(self, in_channels, out_channels, kernel_size, rngs, temporal_kernel=3, dtype=torch.bfloat16,
                 param_dtype=torch.float32):
        super().__init__()
        self.conv = ManualSpectralNorm(Conv(in_channels, out_channels, (temporal_kernel, kernel_size, kernel_size), rngs, dtype,
                                            param_dtype), rngs)
        self.norm = GroupNorm(min(8, out_channels), out_channels, param_dtype)

    def forward(self, x, update_stats=True):
        layer = self.conv.layer
        y, stats = ops.conv3d_with_gn_stats(x.to(layer.dtype), self.conv.normalised_kernel(update_stats), layer.bias,
                                            self.norm.num_groups, None)
        return ops.group_norm_silu(y, self.norm.scale, self.norm.bias, self.norm.num_groups, 1e-6, None, stats)


class DownBlock3D(nn.Module):
    """Two blocks then the spatial max-pool; no skip output.  classifier.py:95-113."""

    def __init__(self, in_channels, out_channels, rngs, temporal_kernel=3, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.conv1 = ConvBlock3D(in_channels, out_channels, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.conv2 = ConvBlock3D(out_channels, out_channels, 3, rngs, temporal_kernel, dtype, param_dtype)

    def forward(self, x, update_stats=True):
        return ops.max_pool_1x2x2(self.conv2(self.conv1(x, update_stats), update_stats))


class Classifier(nn.Module):
    """classifier.py:116-179.  Any clip length; (b, t, h, w, c) -> (b, 1)."""

    def __init__(self, channels, base_features=32, num_levels=4, rngs=None, temporal_kernel=3, dtype=torch.bfloat16,
                 param_dtype=torch.float32):
        super().__init__()
        rngs = rngs if rngs is not None else Rngs(0)
        self.num_levels, self.dtype = num_levels, dtype
        self.initial_conv = ConvBlock3D(channels, base_features, 7, rngs, temporal_kernel, dtype, param_dtype)
        self.encoders = nn.ModuleList()
        in_ch = base_features
        for i in range(num_levels):
            out_ch = base_features * 2 ** (i + 1)
            self.encoders.append(DownBlock3D(in_ch, out_ch, rngs, temporal_kernel, dtype, param_dtype))
            in_ch = out_ch
        self.classifier = _Linear(in_ch, 1, rngs, dtype, param_dtype)

    def forward(self, x, mask=None, train=True, update_stats=True):
        x = self.initial_conv(x.to(self.dtype), update_stats)
        for enc in self.encoders:
            x = enc(x, update_stats)
        x = x.float().mean(dim=(1, 2, 3)).to(self.dtype)             # global average over time, height, width
        return self.classifier(x)


class _Linear(nn.Module):
    """nnx.Linear with the Flax (in, out) kernel layout and names."""

    def __init__(self, in_features, out_features, rngs, dtype, param_dtype):
        super().__init__()
        self.kernel = nn.Parameter(truncated_normal_((in_features, out_features), in_features, rngs.params()).to(param_dtype))
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=param_dtype))
        self.dtype = dtype

    def forward(self, x):
        return F.linear(x.to(self.dtype), self.kernel.to(self.dtype).t(), self.bias.to(self.dtype))
