"""3D-conv UNet with the reference's class surface (train/unet.py), running on the HIP kernels.

Layout is (b, t, h, w, c) as in the reference; every conv / norm / pool / up-conv is a launch of
libvvae_hip.so through ``ops``.  Parameters keep Flax names and layouts (``conv.kernel`` is
(kt, kh, kw, Cin, Cout), ``norm.scale``/``norm.bias``) so checkpoints and the oracle share a key space.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .rngs import Rngs, truncated_normal_


class Conv(nn.Module):
    """nnx.Conv(padding='SAME'), kernel (kt,kh,kw,Cin,Cout), lecun_normal init, zero bias."""

    def __init__(self, in_features, out_features, kernel_size, rngs, dtype=torch.bfloat16, param_dtype=torch.float32,
                 zero_init=False):
        super().__init__()
        kt, kh, kw = kernel_size
        shape = (kt, kh, kw, in_features, out_features)
        k = torch.zeros(shape) if zero_init else truncated_normal_(shape, kt * kh * kw * in_features, rngs.params())
        self.kernel = nn.Parameter(k.to(param_dtype))
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=param_dtype))
        self.dtype = dtype

    def forward(self, x):
        return ops.conv3d(x.to(self.dtype), self.kernel, self.bias)


class ConvTranspose(nn.Module):
    """nnx.ConvTranspose(kernel (1,2,2), strides (1,2,2)), kernel (1,2,2,Cin,Cout)."""

    def __init__(self, in_features, out_features, rngs, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        shape = (1, 2, 2, in_features, out_features)
        self.kernel = nn.Parameter(truncated_normal_(shape, 4 * in_features, rngs.params()).to(param_dtype))
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=param_dtype))
        self.dtype = dtype

    def forward(self, x, out=None, pack=None):
        return ops.conv_transpose_1x2x2(x.to(self.dtype), self.kernel, self.bias, out, pack)


class GroupNorm(nn.Module):
    """nnx.GroupNorm(num_groups, C, eps=1e-6); applied fused with SiLU by ConvBlock3D."""

    def __init__(self, num_groups, num_features, param_dtype=torch.float32):
        super().__init__()
        self.num_groups = num_groups
        self.scale = nn.Parameter(torch.ones(num_features, dtype=param_dtype))
        self.bias = nn.Parameter(torch.zeros(num_features, dtype=param_dtype))


class ConvBlock3D(nn.Module):
    """Conv(kt,k,k) SAME -> GroupNorm(min(8,C)) -> SiLU.  Reference train/unet.py:7-30."""

    def __init__(self, in_channels, out_channels, kernel_size, rngs, temporal_kernel=3,
                 dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.conv = Conv(in_channels, out_channels, (temporal_kernel, kernel_size, kernel_size), rngs, dtype, param_dtype)
        self.norm = GroupNorm(min(8, out_channels), out_channels, param_dtype)

    def forward(self, x, kernel=None, out=None, pack=None, x2=None, price=None, pool=False):
        # ``pool``: -> (y, max_pool(1,2,2)(y)), the pool from the GroupNorm + SiLU launch where ops.gn_silu_pool_ok, else a launch of its own
        # ``price``: (Cin, Cout) the layer really has when ``kernel`` is a zero-padded stand-in (roofline accounting only)
        # ``kernel``: optional stand-in for self.conv.kernel (UNet passes a zero-padded view for 16-channel alignment)
        # ``out``: channel slice of a wider buffer for the block's output (the skip half of a decoder's concat buffer)
        # ``pack``: this step's packed weights of the conv (ops.conv3d_prepack), or None
        # ``x2``: the input is concat([x, x2], channels) held as two tensors (ops.conv3d_cat2_ok was checked by the caller)
        if x2 is not None:
            x, stats = ops.conv3d_cat2_with_gn_stats(x.to(self.conv.dtype), x2.to(self.conv.dtype), self.conv.kernel, self.conv.bias,
                                                     self.norm.num_groups, pack)
        else:
            x, stats = ops.conv3d_with_gn_stats(x.to(self.conv.dtype), self.conv.kernel if kernel is None else kernel, self.conv.bias,
                                                self.norm.num_groups, pack, price)
        if pool and ops.gn_silu_pool_ok(x, self.norm.num_groups, out):
            return ops.group_norm_silu(x, self.norm.scale, self.norm.bias, self.norm.num_groups, 1e-6, out, stats, pool=True)
        y = ops.group_norm_silu(x, self.norm.scale, self.norm.bias, self.norm.num_groups, 1e-6, out, stats)
        if pool:
            pooled, y = ops.max_pool_fork(y)
            return y, pooled
        return y


class DownBlock3D(nn.Module):
    """Two ConvBlock3D then spatial max-pool; returns (pooled, skip).  Reference train/unet.py:33-51."""

    def __init__(self, in_channels, out_channels, rngs, temporal_kernel=3, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.conv1 = ConvBlock3D(in_channels, out_channels, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.conv2 = ConvBlock3D(out_channels, out_channels, 3, rngs, temporal_kernel, dtype, param_dtype)

    def forward(self, x, kernel1=None, skip_out=None, packs=(None, None)):
        price = None if kernel1 is None else tuple(self.conv1.conv.kernel.shape[-2:])
        skip, pooled = self.conv2(self.conv1(x, kernel1, pack=packs[0], price=price), out=skip_out, pack=packs[1], pool=True)
        return pooled, skip


class UpBlock3D(nn.Module):
    """ConvTranspose 2x spatial upsample, concat skip, two ConvBlock3D.  Reference train/unet.py:54-83."""

    def __init__(self, in_channels, out_channels, rngs, temporal_kernel=3, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.upsample = ConvTranspose(in_channels, out_channels, rngs, dtype, param_dtype)
        self.conv1 = ConvBlock3D(out_channels * 2, out_channels, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.conv2 = ConvBlock3D(out_channels, out_channels, 3, rngs, temporal_kernel, dtype, param_dtype)

    def forward(self, x, skip, joint=None, packs=(None, None), up_pack=None):
        """joint: the (.., 2C) buffer whose upper channel half ``skip`` already is; the up-conv writes the lower half and the
        concat of the reference (unet.py:79, a 268 MB copy at the 256^2 level) disappears.  up_pack: the up-conv's packed weights."""
        if joint is not None:
            c = skip.shape[-1]
            x = ops.join_channels(self.upsample(x, out=joint[..., :c], pack=up_pack), skip, joint)
            return self.conv2(self.conv1(x, pack=packs[0]), pack=packs[1])
        up = self.upsample(x, pack=up_pack)
        if ops.conv3d_cat2_ok(up, skip, self.conv1.conv.kernel):
            # 16 + 16 channels: two dense tensors instead of the halves of a joint buffer (a 32-byte half-voxel write runs at a third
            # of the HBM rate); conv1 reads both, its input gradient writes both
            return self.conv2(self.conv1(up, pack=packs[0], x2=skip), pack=packs[1])
        return self.conv2(self.conv1(torch.cat([up, skip], dim=-1), pack=packs[0]), pack=packs[1])


class UNet(nn.Module):
    """Reference train/unet.py:86-188: 3x7x7 patch_mixer, num_levels down, 2 bottleneck blocks, num_levels up,
    zero-initialised 1x1x1 final_conv.  Input/output (b, t, h, w, c)."""

    def __init__(self, channels, base_features=32, num_levels=3, out_features=3, rngs=None, temporal_kernel=3,
                 dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        rngs = rngs if rngs is not None else Rngs(0)
        self.num_levels = num_levels
        self.dtype = dtype
        self.patch_mixer = Conv(channels, channels, (temporal_kernel, 7, 7), rngs, dtype, param_dtype)
        self.encoders = nn.ModuleList()
        in_ch = channels
        for i in range(num_levels):
            out_ch = base_features * (2 ** i)
            self.encoders.append(DownBlock3D(in_ch, out_ch, rngs, temporal_kernel, dtype, param_dtype))
            in_ch = out_ch
        bottleneck_ch = base_features * (2 ** num_levels)
        self.bottleneck1 = ConvBlock3D(in_ch, bottleneck_ch, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.bottleneck2 = ConvBlock3D(bottleneck_ch, bottleneck_ch, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.decoders = nn.ModuleList()
        in_ch = bottleneck_ch
        for i in range(num_levels - 1, -1, -1):
            out_ch = base_features * (2 ** i)
            self.decoders.append(UpBlock3D(in_ch, out_ch, rngs, temporal_kernel, dtype, param_dtype))
            in_ch = out_ch
        self.final_conv = Conv(base_features, out_features, (1, 1, 1), rngs, dtype, param_dtype, zero_init=True)

    def pad_plan(self, on_gpu):
        """-> (parameters, padded sizes of their last dims) for ops.pad_last2_group, or None: the weights forward() zero-pads to the 16-channel
        granule of the matrix-core kernels.  A caller with pads of its own (model.Decoder: the un-embedding's down-projection) runs them all in
        ONE launch each way and hands the padded tensors back as forward(padded=...)."""
        c = self.patch_mixer.kernel.shape[-2]
        pad = (-c) % 16 if (on_gpu and self.dtype == torch.bfloat16 and len(self.encoders) > 0) else 0
        if not pad:
            return None
        km, bm, k1 = self.patch_mixer.kernel, self.patch_mixer.bias, self.encoders[0].conv1.conv.kernel
        if not all(p.dtype == torch.float32 for p in (km, bm, k1)):
            return None
        return [km, bm, k1], [(c + pad, c + pad), (c + pad,), (c + pad, k1.shape[-1])]

    def forward(self, x, residual=None, padded=None):
        """``residual``: (b, t, h, w, out_features) added to the result -- the decoder's ``x + self.unet(feat)`` (reference train/model.py:97)
        inside the final 1x1x1 product where the pointwise kernels take the shape, else a separate add.  ``padded``: the tensors of pad_plan(),
        already padded by the caller."""
        x = x.to(self.dtype)
        c = self.patch_mixer.kernel.shape[-2]
        pad = (-c) % 16 if (x.is_cuda and self.dtype == torch.bfloat16 and len(self.encoders) > 0) else 0
        k1 = None
        km, bm = self.patch_mixer.kernel, self.patch_mixer.bias
        if pad and padded is not None:
            if x.shape[-1] != c + pad:
                raise ValueError(f"UNet expects {c + pad} zero-padded input channels beside padded weights, got {x.shape[-1]}")
            km, bm, k1 = padded
        elif pad:
            # bf16 MFMA kernels want channel counts in multiples of 16: run the mixer and the first encoder conv on
            # zero-padded channels (zero weights in the pad rows/columns => identical results, grads sliced by autograd).
            # A caller may hand the features over already padded (layers.PatchUnEmbedding.forward_padded).
            if x.shape[-1] == c:
                x = F.pad(x, (0, pad))
            elif x.shape[-1] != c + pad:
                raise ValueError(f"UNet expects {c} (or {c + pad} zero-padded) input channels, got {x.shape[-1]}")
            k1 = self.encoders[0].conv1.conv.kernel
            if x.is_cuda and all(p.dtype == torch.float32 for p in (km, bm, k1)):
                # the three pads in one launch (and their backward slices in one): six + four framework launches per step otherwise
                km, bm, k1 = ops.pad_last2_group([km, bm, k1], [(c + pad, c + pad), (c + pad,), (c + pad, k1.shape[-1])])
            else:
                km, bm = F.pad(km, (0, pad, 0, pad)), F.pad(bm, (0, pad))
                k1 = F.pad(k1, (0, 0, 0, pad))
        # the padded mixer tells the kernels how many of its 16 K channels are real: the product over the padding is skipped
        mixer_real = (c, c) if pad else None
        # every conv layer's weights are packed for the matrix-core kernels once per step, in one launch (they were 28 launches)
        packs, upacks = None, [None] * len(self.decoders)
        if x.is_cuda and self.dtype == torch.bfloat16:
            ks = [km]
            for i, enc in enumerate(self.encoders):
                ks += [k1 if (i == 0 and k1 is not None) else enc.conv1.conv.kernel, enc.conv2.conv.kernel]
            ks += [self.bottleneck1.conv.kernel, self.bottleneck2.conv.kernel]
            for dec in self.decoders:
                ks += [dec.conv1.conv.kernel, dec.conv2.conv.kernel]
            packs = ops.conv3d_prepack([k.detach() for k in ks], [mixer_real] + [None] * (len(ks) - 1))
            upacks = ops.convt_prepack([dec.upsample.kernel.detach() for dec in self.decoders])      # likewise the up-convs: six launches
        pk = (lambda j: packs[j]) if packs is not None else (lambda j: None)
        x = ops.conv3d(x.to(self.patch_mixer.dtype), km, bm, pack=pk(0), real=mixer_real)
        skips, joints = [], []
        for i, enc in enumerate(self.encoders):
            c = enc.conv2.norm.scale.shape[0]
            joint = None
            dec_k = self.decoders[len(self.encoders) - 1 - i].conv1.conv.kernel
            two_tensors = (x.is_cuda and self.dtype == torch.bfloat16 and not ops._FORCE_GENERIC[0]
                           and ops.lib().vvae_conv3d_cat2_supported(2 * c, dec_k.shape[-1], c, *dec_k.shape[:3]) == 1)
            if x.is_cuda and x.shape[-3] % 2 == 0 and x.shape[-2] % 2 == 0 and not two_tensors:
                # the decoder at this level will read concat([up, skip]): allocate that buffer now and let the encoder's
                # last kernel write the skip straight into its upper channel half.  (Not at 16 + 16 channels: there the skip and
                # the up-conv output stay two dense tensors and the decoder's conv reads both, see UpBlock3D.)
                joint = torch.empty((*x.shape[:-1], 2 * c), dtype=self.dtype, device=x.device)
            x, skip = enc(x, k1 if i == 0 else None, None if joint is None else joint[..., c:], packs=(pk(1 + 2 * i), pk(2 + 2 * i)))
            skips.append(skip)
            joints.append(joint)
        nb = 1 + 2 * len(self.encoders)
        x = self.bottleneck2(self.bottleneck1(x, pack=pk(nb)), pack=pk(nb + 1))
        for j, (dec, skip, joint) in enumerate(zip(self.decoders, reversed(skips), reversed(joints))):
            x = dec(x, skip, joint, packs=(pk(nb + 2 + 2 * j), pk(nb + 3 + 2 * j)), up_pack=upacks[j])
        if residual is None:
            return self.final_conv(x)
        fc = self.final_conv
        x = x.to(fc.dtype)
        residual = residual.to(fc.dtype)
        if ops.conv3d_pointwise_add_ok(x, fc.kernel, residual):
            return ops.conv3d_pointwise_add(x, fc.kernel, fc.bias, residual)
        return residual + fc(x)
