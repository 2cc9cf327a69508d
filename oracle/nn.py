"""Flax-NNX primitive semantics restated in CPU PyTorch (test infrastructure only).

Layout everywhere is channels-last ``(b, t, h, w, c)`` as in the reference
(``train/unet.py:26``).  Parameters are kept in Flax layout: Conv kernel
``(kt, kh, kw, Cin, Cout)``, Linear kernel ``(in, out)``.

``dtype`` emulates Flax's ``dtype=`` (compute dtype): inputs, kernel and bias
are rounded to ``dtype`` before the op, the op accumulates in fp32, the result
is rounded to ``dtype``.  All tensors stay fp32
containers so every op runs on the CPU fp32 kernels.
"""
import math

import torch
import torch.nn.functional as F

F32 = torch.float32


def q(x, dtype):
    """Round ``x`` to ``dtype`` and return it as fp32 (promote_dtype emulation)."""
    if dtype is None or dtype == F32:
        return x.to(F32)
    return x.to(dtype).to(F32)


def linear(x, kernel, bias, dtype=F32):
    """nnx.Linear: ``y = x @ kernel + bias`` with kernel (in, out).  layers.py:15,142."""
    y = q(x, dtype) @ q(kernel, dtype)
    y = q(y, dtype)
    if bias is not None:
        y = q(y + q(bias, dtype), dtype)
    return y


def layer_norm(x, scale, bias, dtype=F32, eps=1e-6):
    """nnx.LayerNorm(eps=1e-6, use_fast_variance=True) over the last axis.

    layers.py:17,152,155-156,178.  Stats in fp32: var = max(0, E[x^2]-E[x]^2).
    """
    x = q(x, dtype)
    mean = x.mean(-1, keepdim=True)
    mean2 = (x * x).mean(-1, keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    mul = torch.rsqrt(var + eps)
    if scale is not None:
        mul = mul * scale.to(F32)
    y = (x - mean) * mul
    if bias is not None:
        y = y + bias.to(F32)
    return q(y, dtype)


def group_norm(x, scale, bias, num_groups, dtype=F32, eps=1e-6):
    """nnx.GroupNorm(num_groups, C, eps=1e-6) on (b,t,h,w,c).  unet.py:22-23.

    Statistics per (sample, group) over (t, h, w, c/G), fp32, fast variance.
    """
    x = q(x, dtype)
    b = x.shape[0]
    c = x.shape[-1]
    xg = x.reshape(b, -1, num_groups, c // num_groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    mean2 = (xg * xg).mean(dim=(1, 3), keepdim=True)
    var = torch.clamp(mean2 - mean * mean, min=0.0)
    rstd = torch.rsqrt(var + eps)
    y = ((xg - mean) * rstd).reshape(x.shape)
    y = y * scale.to(F32) + bias.to(F32)
    return q(y, dtype)


def silu(x, dtype=F32):
    """nnx.silu = x * sigmoid(x).  unet.py:29."""
    x = q(x, dtype)
    return q(x * torch.sigmoid(x), dtype)


def softplus(x):
    """jax.nn.softplus = logaddexp(x, 0).  model.py:54."""
    return torch.logaddexp(x, torch.zeros((), dtype=x.dtype))


def conv3d_same(x, kernel, bias, dtype=F32):
    """nnx.Conv(padding='SAME', stride 1): cross-correlation, zero pad (k-1)/2.

    unet.py:13-21,111-113,144-153.  kernel (kt,kh,kw,Cin,Cout) ->
    torch weight[o,i,t,h,w] = kernel[t,h,w,i,o].
    """
    kt, kh, kw, _, _ = kernel.shape
    assert kt % 2 == 1 and kh % 2 == 1 and kw % 2 == 1
    xn = q(x, dtype).permute(0, 4, 1, 2, 3)
    w = q(kernel, dtype).permute(4, 3, 0, 1, 2)
    y = F.conv3d(xn, w, bias=None, padding=(kt // 2, kh // 2, kw // 2))
    y = q(y.permute(0, 2, 3, 4, 1), dtype)
    if bias is not None:
        y = q(y + q(bias, dtype), dtype)
    return y


def conv_transpose_1x2x2(x, kernel, bias, dtype=F32):
    """nnx.ConvTranspose(kernel (1,2,2), strides (1,2,2), padding 'SAME').

    unet.py:61-69.  lax.conv_transpose with transpose_kernel=False is an
    lhs-dilated correlation with the kernel *not* flipped; for k=s=2 the SAME
    padding is (1,1) so ``out[2i+d] = x[i] * K[1-d]`` per spatial axis
     -- spatially flipped relative to torch ConvTranspose3d:
    W_torch[i,o,0,a,b] = K[0,1-a,1-b,i,o].
    """
    assert tuple(kernel.shape[:3]) == (1, 2, 2)
    xn = q(x, dtype).permute(0, 4, 1, 2, 3)
    k = q(kernel, dtype)
    w = torch.flip(k, dims=(1, 2)).permute(3, 4, 0, 1, 2).contiguous()
    y = F.conv_transpose3d(xn, w, bias=None, stride=(1, 2, 2))
    y = q(y.permute(0, 2, 3, 4, 1), dtype)
    if bias is not None:
        y = q(y + q(bias, dtype), dtype)
    return y


def conv_transpose_1x2x2_explicit(x, kernel, bias):
    """Same op, derived literally: dilate lhs by the stride, pad (1,1), correlate.

    Used only by the self-consistency test that guards the flip convention.
    """
    b, t, h, w, ci = x.shape
    co = kernel.shape[-1]
    dil = torch.zeros(b, t, 2 * h - 1, 2 * w - 1, ci, dtype=x.dtype)
    dil[:, :, ::2, ::2, :] = x
    pad = F.pad(dil, (0, 0, 1, 1, 1, 1))
    out = torch.zeros(b, t, 2 * h, 2 * w, co, dtype=x.dtype)
    for a in range(2):
        for c in range(2):
            out += pad[:, :, a:a + 2 * h, c:c + 2 * w, :] @ kernel[0, a, c]
    return out + bias


def max_pool_1x2x2(x):
    """nnx.max_pool(x, (1,2,2), strides=(1,2,2)), padding VALID.  unet.py:50."""
    xn = x.permute(0, 4, 1, 2, 3)
    y = F.max_pool3d(xn, kernel_size=(1, 2, 2), stride=(1, 2, 2))
    return y.permute(0, 2, 3, 4, 1)


def lecun_normal_(shape, fan_in, gen, scale=1.0):
    """variance_scaling(scale, 'fan_in', 'truncated_normal')."""
    std = math.sqrt(scale / fan_in) / 0.87962566103423978
    t = torch.empty(shape, dtype=F32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return t * std
