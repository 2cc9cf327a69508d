"""CPU restatement of the reference 3D-conv UNet (train/unet.py).  Test infrastructure only.

Parameters arrive as a flat ``{dotted.name: tensor}`` dict whose names mirror
the reference's attribute paths (``encoders.0.conv1.conv.kernel`` ...).
"""
import torch

from . import nn as O


def sub(p, prefix):
    """View of the flat dict under ``prefix.`` with the prefix stripped."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix + ".")}


def conv_block3d(p, x, dtype=O.F32):
    """ConvBlock3D.__call__: conv -> GroupNorm(min(8,C)) -> SiLU.  unet.py:26-30."""
    cout = p["conv.kernel"].shape[-1]
    x = O.conv3d_same(x, p["conv.kernel"], p["conv.bias"], dtype)
    x = O.group_norm(x, p["norm.scale"], p["norm.bias"], min(8, cout), dtype)
    return O.silu(x, dtype)


def down_block3d(p, x, dtype=O.F32):
    """DownBlock3D.__call__: conv1, conv2, skip, spatial max-pool.  unet.py:45-51."""
    x = conv_block3d(sub(p, "conv1"), x, dtype)
    x = conv_block3d(sub(p, "conv2"), x, dtype)
    skip = x
    return O.max_pool_1x2x2(x), skip


def up_block3d(p, x, skip, dtype=O.F32):
    """UpBlock3D.__call__: upsample, concat(x, skip), conv1, conv2.  unet.py:77-83."""
    x = O.conv_transpose_1x2x2(x, p["upsample.kernel"], p["upsample.bias"], dtype)
    x = torch.cat([x, skip], dim=-1)
    x = conv_block3d(sub(p, "conv1"), x, dtype)
    return conv_block3d(sub(p, "conv2"), x, dtype)


def unet_num_levels(p):
    n = 0
    while f"encoders.{n}.conv1.conv.kernel" in p:
        n += 1
    return n


def unet(p, x, dtype=O.F32):
    """UNet.__call__.  unet.py:155-188."""
    x = O.q(x, dtype)                                                     # :166
    x = O.conv3d_same(x, p["patch_mixer.kernel"], p["patch_mixer.bias"], dtype)  # :169
    levels = unet_num_levels(p)
    skips = []
    for i in range(levels):                                               # :173-175
        x, s = down_block3d(sub(p, f"encoders.{i}"), x, dtype)
        skips.append(s)
    x = conv_block3d(sub(p, "bottleneck1"), x, dtype)                     # :178
    x = conv_block3d(sub(p, "bottleneck2"), x, dtype)                     # :179
    for i, s in zip(range(levels), reversed(skips)):                      # :182-183
        x = up_block3d(sub(p, f"decoders.{i}"), x, s, dtype)
    return O.conv3d_same(x, p["final_conv.kernel"], p["final_conv.bias"], dtype)  # :186


def init_conv_block(p, prefix, cin, cout, k, kt, gen):
    p[f"{prefix}.conv.kernel"] = O.lecun_normal_((kt, k, k, cin, cout), kt * k * k * cin, gen)
    p[f"{prefix}.conv.bias"] = torch.zeros(cout)
    p[f"{prefix}.norm.scale"] = torch.ones(cout)
    p[f"{prefix}.norm.bias"] = torch.zeros(cout)


def init_unet(channels, base_features=32, num_levels=3, out_features=3, seed=0,
              temporal_kernel=3, zero_final=True):
    """Parameter tree of UNet.__init__ (unet.py:93-153) with Flax default inits.

    ``zero_final=False`` re-initialises final_conv non-zero (parity/bench use;
    the reference's zero init makes the UNet a no-op, unet.py:150).
    """
    gen = torch.Generator().manual_seed(seed)
    kt = temporal_kernel
    p = {}
    p["patch_mixer.kernel"] = O.lecun_normal_((kt, 7, 7, channels, channels), kt * 49 * channels, gen)
    p["patch_mixer.bias"] = torch.zeros(channels)
    cin = channels
    for i in range(num_levels):
        cout = base_features * 2 ** i
        init_conv_block(p, f"encoders.{i}.conv1", cin, cout, 3, kt, gen)
        init_conv_block(p, f"encoders.{i}.conv2", cout, cout, 3, kt, gen)
        cin = cout
    bc = base_features * 2 ** num_levels
    init_conv_block(p, "bottleneck1", cin, bc, 3, kt, gen)
    init_conv_block(p, "bottleneck2", bc, bc, 3, kt, gen)
    cin = bc
    for j, i in enumerate(range(num_levels - 1, -1, -1)):
        cout = base_features * 2 ** i
        p[f"decoders.{j}.upsample.kernel"] = O.lecun_normal_((1, 2, 2, cin, cout), 4 * cin, gen)
        p[f"decoders.{j}.upsample.bias"] = torch.zeros(cout)
        init_conv_block(p, f"decoders.{j}.conv1", 2 * cout, cout, 3, kt, gen)
        init_conv_block(p, f"decoders.{j}.conv2", cout, cout, 3, kt, gen)
        cin = cout
    if zero_final:
        p["final_conv.kernel"] = torch.zeros(1, 1, 1, base_features, out_features)
    else:
        p["final_conv.kernel"] = O.lecun_normal_((1, 1, 1, base_features, out_features), base_features, gen)
    p["final_conv.bias"] = torch.zeros(out_features)
    return p
