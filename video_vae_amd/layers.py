"""Transformer-side layers with the reference's class surface (train/layers.py).

Everything on the bf16 GPU path is a launch of libvvae_hip.so through ``ops``: LayerNorm (with the residual adds folded in), the
temporal AND the spatial attention cores (q/k-norm + RoPE + softmax + PV in one kernel each: ``ops.temporal_attention_core``,
``ops.spatial_attention_core``), fc1 + SiLU and the MLP's input gradient on the own NT GEMM, every weight gradient on the grouped
TN GEMM.  The remaining forward / input-gradient products of the Linear layers are plain hipBLASLt GEMMs.  The library's
flash-attention kernel is only the fallback core for spatial shapes the fused kernel does not take (head_dim != 64, S > 256).
"""
import math

import torch
import torch.nn.functional as F
from einops import rearrange
from torch import nn

from . import ops
from .rngs import truncated_normal_


_MM_F32_OUT = [None]     # does torch.mm(bf16, bf16, out_dtype=float32) work on this build? decided on first use


def _mm_f32(a, b):
    """a @ b for bf16 operands with an fp32 result (weight gradients go straight to fp32, no bf16 round trip)."""
    if _MM_F32_OUT[0] is None:
        try:
            torch.mm(a[:1], b[:, :1], out_dtype=torch.float32)
            _MM_F32_OUT[0] = True
        except Exception:
            _MM_F32_OUT[0] = False
    if _MM_F32_OUT[0]:
        return torch.mm(a, b, out_dtype=torch.float32)
    return torch.mm(a, b).float()


_BMM_F32_OUT = [None]


FRAMEWORK_COLSUM = [False]     # tools/reduce_history_probe.py only: put the framework's multi-block reduction back to study it


def _colsum_f32(dy2):
    """Bias gradient dy2.sum(0) in fp32.  On the GPU through the two-stage HIP reduction (vvae_colsum): the framework's multi-block
    reduction gave history-dependent results inside a replayed hipGraph (tools/step_determinism.py, DESIGN.md section 3: its semaphore
    reset is a hipMemsetAsync, i.e. a memset NODE in the captured graph) -- stale values, occasionally garbage large enough to end a
    run in NaN."""
    if dy2.is_cuda and dy2.dtype in (torch.bfloat16, torch.float32) and dy2.stride(-1) == 1 and not FRAMEWORK_COLSUM[0]:
        return ops.colsum_raw(dy2)
    return dy2.sum(0, dtype=torch.float32)


def _dw_f32(x2, dy2):
    """x2^T @ dy2 (K tokens x M, K x N -> M x N fp32) for the Linear layers the HIP weight-gradient kernels do not take
    (M or N not a multiple of 128: patch embedding / un-embedding, the latent heads).  The library picks 64x64 tiles for such
    products, i.e. a few dozen workgroups each walking all 16 384 tokens; splitting the tokens into independent batches first
    (a batched product, then a fixed-order sum of the partial results) fills the chip: ~97 us -> ~20 us per product."""
    k, m = x2.shape
    n = dy2.shape[1]
    tiles = -(-m // 64) * -(-n // 64)
    split = 1
    while split < 32 and tiles * split < 512 and k % (2 * split) == 0 and k // (2 * split) >= 512:
        split *= 2
    if split > 1 and x2.is_contiguous() and dy2.is_contiguous():
        if _BMM_F32_OUT[0] is None:
            try:
                torch.bmm(x2[:2, :1].reshape(1, 1, 2), dy2[:2, :1].reshape(1, 2, 1), out_dtype=torch.float32)
                _BMM_F32_OUT[0] = True
            except Exception:
                _BMM_F32_OUT[0] = False
        if _BMM_F32_OUT[0]:
            part = torch.bmm(x2.view(split, k // split, m).transpose(1, 2), dy2.view(split, k // split, n), out_dtype=torch.float32)
            return part.sum(0)
    return _mm_f32(x2.t(), dy2)


_MASK_U8 = [None]          # (mask tensor, its version, t, uint8 form): every block of a step is handed the same mask object


def _mask_u8(mask, t):
    """(rows, t) uint8 form of the temporal mask, converted once per mask tensor instead of once per attention block."""
    hit = _MASK_U8[0]
    if hit is not None and hit[0] is mask and hit[1] == mask._version and hit[2] == t:
        return hit[3]
    m8 = mask.reshape(-1, t).to(torch.uint8).contiguous()
    _MASK_U8[0] = (mask, mask._version, t, m8)
    return m8


class _LinearBf16(torch.autograd.Function):
    """y = x @ W + b in bf16 on hipBLASLt with fp32 master weights: the forward reads the optimizer's bf16 shadow copy
    (``param.bf16``, refreshed by the fused Adam kernel) so no per-call weight cast runs, and the backward returns
    fp32 weight / bias gradients directly."""

    @staticmethod
    def forward(ctx, x, kernel, bias, res=None, with_silu=False):
        """``res``: the residual stream this Linear closes a branch of (x_skip + Linear(...)): added inside the product (the
        library's, with the residual as its C operand: ops.linear_residual; in situ it beats the own NT kernel's residual epilogue,
        28.6 vs 34 us on the out-projection); the caller checked ops.linear_residual_ok.  ``with_silu``: also return silu(y) (the caller checked nt_silu_ok)."""
        wb = getattr(kernel, "bf16", None)
        if wb is None:
            wb = kernel.detach().to(torch.bfloat16)
        bb = getattr(bias, "bf16", None)
        if bb is None:
            bb = bias.detach().to(torch.bfloat16)
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, wb)
        ctx.xshape = x.shape
        ctx.kparam, ctx.bparam = kernel, bias            # for ops.deferred_wgrad: where the parked gradient is to be written
        ctx.has_res = res is not None
        ctx.two = bool(with_silu)
        wt = getattr(kernel, "bf16_t", None)                         # (out, in) shadow: the own NT GEMM's operand (optim.Optimizer)
        wl = wt if WT_LIBRARY else None
        own = wt is not None and bias.dtype == torch.float32 and x2.stride(-1) == 1 and ops.gemm_nt_supported(x2, wt)
        if with_silu:
            # -> (h, silu(h)) from ONE product (ops.gemm_nt, EPI_SILU): the activation between the MLP's Linear layers costs no pass
            a, h = ops.gemm_nt(x2, wt, bias.detach(), None, ops.EPI_SILU)
            h, a = h.view(*x.shape[:-1], wb.shape[1]), a.view(*x.shape[:-1], wb.shape[1])
            ctx.mark_non_differentiable(a)
            ctx.set_materialize_grads(False)                        # no zero-filled gradient tensor for ``a`` in backward
            return h, a
        if res is not None:
            if own and (OWN & 2):                                      # the add in the own product's epilogue (on the rounded Linear output, as the reference adds)
                return ops.gemm_nt(x2, wt, bias.detach(), res.reshape(-1, wb.shape[1]), ops.EPI_RES).view(res.shape)
            return ops.linear_residual(x2, wb, bb, res.reshape(-1, wb.shape[1]), wl).view(res.shape)
        if own and (OWN & 1):
            return ops.gemm_nt(x2, wt, bias.detach()).view(*x.shape[:-1], wb.shape[1])
        if wl is not None:                                           # the library's K-contiguous-both-sides kernel: 8 % faster on qkv
            return torch.addmm(bb, x2, wl.t()).view(*x.shape[:-1], wb.shape[1])
        return torch.addmm(bb, x2, wb).view(*x.shape[:-1], wb.shape[1])

    @staticmethod
    def backward(ctx, dy, da=None):
        x2, wb = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != torch.bfloat16:
            dy2 = dy2.to(torch.bfloat16)
        dres = dy if ctx.has_res else None               # the residual edge: identity
        dx = None
        if ctx.needs_input_grad[0]:
            # dy (M, out) . W (in, out)^T: the weight as stored is the K-contiguous (N, K) operand of the own NT GEMM
            dy2c = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
            dx = (ops.gemm_nt(dy2c, wb) if ((OWN & 4) and ops.gemm_nt_supported(dy2c, wb)) else torch.mm(dy2, wb.t())).view(ctx.xshape)
        # parked (multiplied after backward in a grouped launch, straight into the optimizer's flat gradient buffer), the split-K HIP kernel
        # (bias gradient rides along: K = tokens >> M, N), or the batched library product
        dw, db = _linear_param_grads(x2, dy2, ctx.kparam, ctx.bparam, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, dres, None


def _linear_param_grads(x2, dy2, kparam, bparam, need_w, need_b):
    """(dW, db) of a bf16 Linear as _LinearBf16.backward forms them: parked for the grouped launch (-> None, None), the own split-K
    kernel, or the batched library product for widths it does not take."""
    if need_w and need_b and ops.wgrad_deferrable(x2, dy2, kparam, bparam):
        ops.WGRAD_QUEUE[0].append((x2, dy2, kparam, bparam))
        return None, None
    if need_w and ops.gemm_tn_supported(x2, dy2):
        return ops.gemm_tn(x2, dy2, need_b)
    return (_dw_f32(x2, dy2) if need_w else None), (_colsum_f32(dy2) if need_b else None)


class _LinearPairBf16(torch.autograd.Function):
    """(x @ W1 + b1, x @ W2 + b2) for two Linear layers reading the same tensor (the encoder's mean and variance heads, reference
    train/model.py:53-55).  One autograd node so that the two input gradients meet INSIDE the second product (dy2 @ W2^T accumulated onto
    dy1 @ W1^T through the library's C operand) instead of in a 25 MB add launch behind two products."""

    @staticmethod
    def forward(ctx, x, k1, b1, k2, b2):
        x2 = x.reshape(-1, x.shape[-1])
        w = [getattr(k, "bf16", None) if getattr(k, "bf16", None) is not None else k.detach().to(torch.bfloat16) for k in (k1, k2)]
        b = [getattr(t, "bf16", None) if getattr(t, "bf16", None) is not None else t.detach().to(torch.bfloat16) for t in (b1, b2)]
        ctx.save_for_backward(x2, w[0], w[1])
        ctx.xshape = x.shape
        ctx.params = (k1, b1, k2, b2)
        ctx.set_materialize_grads(False)
        return tuple(torch.addmm(bb, x2, wb).view(*x.shape[:-1], wb.shape[1]) for wb, bb in zip(w, b))

    @staticmethod
    def backward(ctx, dy1, dy2):
        x2, w1, w2 = ctx.saved_tensors
        k1, b1, k2, b2 = ctx.params
        g = [None if d is None else d.reshape(-1, d.shape[-1]).to(torch.bfloat16) for d in (dy1, dy2)]
        dx = None
        if ctx.needs_input_grad[0]:
            for d, wb in zip(g, (w1, w2)):
                if d is not None:
                    dx = torch.mm(d, wb.t()) if dx is None else dx.addmm_(d, wb.t())
            dx = dx.view(ctx.xshape) if dx is not None else None
        out = [dx]
        for i, (d, kp, bp) in enumerate(((g[0], k1, b1), (g[1], k2, b2))):
            if d is None:
                out += [None, None]
            else:
                out += list(_linear_param_grads(x2, d, kp, bp, ctx.needs_input_grad[1 + 2 * i], ctx.needs_input_grad[2 + 2 * i]))
        return tuple(out)


def linear_pair(x, lin1, lin2):
    """(lin1(x), lin2(x)) for two Linear modules of equal dtype; on the bf16 GPU path one autograd node (_LinearPairBf16)."""
    x = x.to(lin1.dtype)
    if (lin1.dtype == torch.bfloat16 and lin2.dtype == torch.bfloat16 and x.is_cuda and lin1.kernel.dtype == torch.float32
            and lin2.kernel.dtype == torch.float32):
        return _LinearPairBf16.apply(x, lin1.kernel, lin1.bias, lin2.kernel, lin2.bias)
    return lin1(x), lin2(x)


class _SiluLinearBf16(torch.autograd.Function):
    """linear2(silu(h)) of the MLP (reference train/layers.py:186-189) as one autograd node, so that its backward can form
    dh = (dy @ W2^T) * silu'(h) in the epilogue of ONE product (ops.gemm_nt, EPI_MUL_DSILU: the input gradient never makes the
    round trip through HBM that a separate silu_backward launch costs).  Forward = the library GEMM + the framework's SiLU."""

    @staticmethod
    def forward(ctx, h, kernel, bias, res=None, act=None):
        """``act``: silu(h) when the producing product already made it (_LinearBf16 with_silu)."""
        wb, bb = kernel.bf16, bias.bf16
        h2 = h.reshape(-1, h.shape[-1])
        a = act.reshape(-1, h.shape[-1]) if act is not None else ops.silu_bf16(h2)
        ctx.save_for_backward(h2, a, wb)
        ctx.hshape = h.shape
        ctx.kparam, ctx.bparam = kernel, bias
        ctx.has_res = res is not None
        wt = getattr(kernel, "bf16_t", None) if WT_LIBRARY else None
        wo = getattr(kernel, "bf16_t", None)
        if res is not None and (OWN & 2) and wo is not None and bias.dtype == torch.float32 and ops.gemm_nt_supported(a, wo):
            return ops.gemm_nt(a, wo, bias.detach(), res.reshape(-1, wb.shape[1]), ops.EPI_RES).view(res.shape)
        if res is not None:                                          # x_skip + linear2(silu(h)): the add rides in the library product
            return ops.linear_residual(a, wb, bb, res.reshape(-1, wb.shape[1]), wt).view(res.shape)
        if (OWN & 1) and wo is not None and bias.dtype == torch.float32 and ops.gemm_nt_supported(a, wo):      # the same product Linear itself runs
            return ops.gemm_nt(a, wo, bias.detach()).view(*h.shape[:-1], wb.shape[1])
        return torch.addmm(bb, a, wb if wt is None else wt.t()).view(*h.shape[:-1], wb.shape[1])

    @staticmethod
    def backward(ctx, dy):
        h2, a, wb = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != torch.bfloat16:
            dy2 = dy2.to(torch.bfloat16)
        dy2 = dy2.contiguous()
        dres = dy if ctx.has_res else None
        dh = None
        if ctx.needs_input_grad[0]:
            if ops.gemm_nt_supported(dy2, wb):                      # wb (mlp, out) is the (N, K) operand as stored
                dh = ops.gemm_nt(dy2, wb, None, h2, ops.EPI_MUL_DSILU)
            else:
                dh = torch.ops.aten.silu_backward(torch.mm(dy2, wb.t()), h2)
            dh = dh.view(ctx.hshape)
        dw = db = None
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and ops.wgrad_deferrable(a, dy2, ctx.kparam, ctx.bparam):
            ops.WGRAD_QUEUE[0].append((a, dy2, ctx.kparam, ctx.bparam))
            return dh, None, None, dres, None
        if ctx.needs_input_grad[1] and ops.gemm_tn_supported(a, dy2):
            dw, db = ops.gemm_tn(a, dy2, ctx.needs_input_grad[2])
        else:
            dw = _dw_f32(a, dy2) if ctx.needs_input_grad[1] else None
            db = _colsum_f32(dy2) if ctx.needs_input_grad[2] else None
        return dh, dw, db, dres, None


def _shadowed(linear, h):
    return (linear.dtype == torch.bfloat16 and h.is_cuda and h.dtype == torch.bfloat16 and linear.kernel.dtype == torch.float32
            and getattr(linear.kernel, "bf16", None) is not None and getattr(linear.bias, "bf16", None) is not None)


def silu_linear(h, linear):
    """linear(silu(h)); fused-backward form when the layer runs the bf16 GPU path with shadowed fp32 parameters."""
    if _shadowed(linear, h):
        return _SiluLinearBf16.apply(h, linear.kernel, linear.bias)
    return linear(F.silu(h))


# Which Linear products run on the own NT GEMM (ops.gemm_nt: csrc/gemm_pp.hip) instead of the library -- a bit mask so that tools/ab_hook.py can
# A/B the routes in situ: 1 = plain forward (qkv), 2 = forward + residual (out-projection, fc2), 4 = plain input gradients (fc1, qkv, out-projection).
OWN = 7
NT_SILU = 1            # fc1 + SiLU as one own NT product (0: library product + SiLU stream kernel); a switch for tools/ab_hook.py
WT_LIBRARY = 1         # qkv / fc2 on the library's K-contiguous kernels through the transposed shadows (0: the (in, out) operand)


def nt_silu_ok(linear, x):
    """linear(x) and silu(linear(x)) can come out of one product of the own NT GEMM (needs the (out, in) weight shadow)."""
    wt = getattr(linear.kernel, "bf16_t", None)
    return (NT_SILU and _shadowed(linear, x) and wt is not None and linear.bias.dtype == torch.float32
            and ops.gemm_nt_supported(x.reshape(-1, x.shape[-1]), wt))


def linear_with_silu(linear, x):
    """-> (h, silu(h)) with h = linear(x): one launch when nt_silu_ok, else (h, None) and the consumer applies the activation."""
    if nt_silu_ok(linear, x):
        return _LinearBf16.apply(x.to(linear.dtype), linear.kernel, linear.bias, None, True)
    return linear(x), None


def close_branch(linear, o, skip, defer, silu=False, act=None):
    """The end of a residual branch: ``skip + linear(o)`` (``linear(silu(o))`` with silu=True; reference train/layers.py:212-221).
    On the bf16 GPU path the add rides in the product and the result is the new residual stream:
    -> (sum, None) if defer else sum.  Elsewhere -> (skip, branch) if defer (the next LayerNorm kernel adds) else skip + branch.
    ``act``: silu(o) if the caller already has it (linear_with_silu)."""
    if _shadowed(linear, o) and skip.dtype == torch.bfloat16:
        o2, r2 = o.reshape(-1, o.shape[-1]), skip.reshape(-1, skip.shape[-1])
        if ops.linear_residual_ok(o2, linear.kernel.bf16, linear.bias.bf16, r2):
            s = (_SiluLinearBf16.apply(o, linear.kernel, linear.bias, skip, act) if silu
                 else _LinearBf16.apply(o, linear.kernel, linear.bias, skip))
            return (s, None) if defer else s
    y = (silu_linear(o, linear) if act is None else linear(act)) if silu else linear(o)
    return (skip, y) if defer else skip + y


class Linear(nn.Module):
    """nnx.Linear: y = x @ kernel + bias, kernel (in, out), lecun_normal init (variance_scaling(scale))."""

    def __init__(self, in_features, out_features, rngs, dtype=torch.bfloat16, param_dtype=torch.float32, kernel_scale=1.0):
        super().__init__()
        k = truncated_normal_((in_features, out_features), in_features, rngs.params(), kernel_scale)
        self.kernel = nn.Parameter(k.to(param_dtype))
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=param_dtype))
        self.dtype = dtype
        # (out, in) bf16 shadow kept by the optimizer: the weight operand of the own NT GEMM's forward product (csrc/gemm_pp.hip)
        self.kernel.want_t = in_features % 64 == 0 and out_features % 64 == 0

    def forward(self, x):
        x = x.to(self.dtype)
        if self.dtype == torch.bfloat16 and x.is_cuda and self.kernel.dtype == torch.float32:
            return _LinearBf16.apply(x, self.kernel, self.bias)
        return torch.addmm(self.bias.to(self.dtype), x.reshape(-1, x.shape[-1]), self.kernel.to(self.dtype)).view(
            *x.shape[:-1], self.kernel.shape[1])


class LayerNorm(nn.Module):
    """nnx.LayerNorm(eps=1e-6) over the last axis, fp32 statistics, optional bias."""

    def __init__(self, num_features, dtype=torch.bfloat16, param_dtype=torch.float32, use_bias=True):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(num_features, dtype=param_dtype))
        self.bias = nn.Parameter(torch.zeros(num_features, dtype=param_dtype)) if use_bias else None
        self.dtype = dtype

    def forward(self, x):
        x = x.to(self.dtype)
        if ops.layer_norm_supported(x):
            return ops.layer_norm(x, self.scale, self.bias, 1e-6)          # HIP kernel, fp32 statistics and affine
        if x.is_cuda:
            ops.note_fallback(("layer_norm", x.shape[-1], x.dtype), f"LayerNorm over {x.shape[-1]} features of {x.dtype}: not a shape the HIP kernel "
                              "takes, the framework's layer_norm runs")
        b = self.bias.to(self.dtype) if self.bias is not None else None
        return F.layer_norm(x, (x.shape[-1],), self.scale.to(self.dtype), b, 1e-6)

    def fork(self, x, pending=None):
        """-> (LayerNorm(x), x_skip) for ``x_skip + f(LayerNorm(x))``: on the HIP path x_skip is x routed through the
        LayerNorm node, so the skip gradient is added inside its backward kernel (no separate add launch).

        ``pending = (skip, o)``: the residual stream is still the un-added pair of the previous block; x = skip + o is formed
        inside the LayerNorm kernel (one pass over the stream instead of an add launch followed by a LayerNorm launch)."""
        if pending is not None:
            skip, o = pending
            if o is None:                                # the previous branch already added itself (close_branch)
                x = skip
            elif skip.dtype == self.dtype and ops.layer_norm_supported(skip):
                return ops.add_layer_norm_fork(skip, o, self.scale, self.bias, 1e-6)
            else:
                x = skip + o
        if x.dtype == self.dtype and ops.layer_norm_supported(x):
            return ops.layer_norm_fork(x, self.scale, self.bias, 1e-6)
        return self.forward(x), x


class PatchEmbedding(nn.Module):
    """Reference train/layers.py:8-27: patchify -> LayerNorm -> Linear."""

    def __init__(self, height, width, channels, patch_size, rngs, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.patch_size = patch_size
        self.dtype = dtype
        d = patch_size * patch_size * channels
        self.linear = Linear(d, d, rngs, dtype, param_dtype)
        self.norm = LayerNorm(d, dtype, param_dtype)

    def forward(self, x):
        x = rearrange(x, "b t (h p1) (w p2) c -> b t (h w) (p1 p2 c)", p1=self.patch_size, p2=self.patch_size)
        x = x.to(self.dtype)
        return self.linear(self.norm(x))


class _UnpatchPad(torch.autograd.Function):
    """"b t (h w) (p1 p2 c u) -> b t (h p1) (w p2) (c u)" (reference train/layers.py:48) into a buffer whose channel count is
    rounded up to a multiple of 16, pad channels zero: one HIP launch each way (vvae_unpatch_pad_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, p, h, w, u, pad):
        b, t = x.shape[:2]
        cu = x.shape[-1] // (p * p)
        x = x.contiguous()
        out = torch.empty((b, t, h * p, w * p, cu + pad), dtype=x.dtype, device=x.device)
        ops.unpatch_pad(x, out, b * t, h, w, p, cu, cu + pad, backward=False)
        ctx.dims = (p, h, w, cu, pad, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        p, h, w, cu, pad, shape = ctx.dims
        g = g.contiguous()
        gx = torch.empty(shape, dtype=g.dtype, device=g.device)
        ops.unpatch_pad(g, gx, shape[0] * shape[1], h, w, p, cu, cu + pad, backward=True)
        return gx, None, None, None, None, None


class PatchUnEmbedding(nn.Module):
    """Reference train/layers.py:29-55 -> (conv features (b,t,H,W,c*u), coarse reconstruction (b,t,H,W,c))."""

    def __init__(self, height, width, channels, patch_size, upsample_rate, rngs, dtype=torch.bfloat16,
                 param_dtype=torch.float32):
        super().__init__()
        self.patch_size, self.height, self.width, self.upsample_rate = patch_size, height, width, upsample_rate
        d = patch_size * patch_size * channels
        self.upsample = Linear(d, d * upsample_rate, rngs, dtype, param_dtype)
        self.downsample = Linear(channels * upsample_rate, channels, rngs, dtype, param_dtype)
        self.linear = Linear(d, d, rngs, dtype, param_dtype)

    def forward_padded(self, x, more_pads=None):
        """forward() for the decoder's bf16 GPU path: the features come back with their channels zero-padded to a multiple of 16
        (what the UNet's matrix-core kernels want), written by ONE strided copy (un-patchify + pad) instead of a rearrange copy
        followed by a pad copy, and the 1x1x1 down-projection reads the same buffer through zero-padded weight rows.
        ``more_pads`` = (tensors, sizes) (unet.UNet.pad_plan): padded in the same launch as the down-projection's rows; the result is then
        (feat, coarse, padded tensors)."""
        p, u = self.patch_size, self.upsample_rate
        cu = self.downsample.kernel.shape[0]
        pad = (-cu) % 16
        if not (x.is_cuda and self.upsample.dtype == torch.bfloat16 and self.upsample.kernel.dtype == torch.float32 and pad and cu % 4 == 0):
            return None                                  # the caller falls back to forward() (the fused kernel moves 8-byte pieces)
        x = self.upsample(self.linear(x))
        feat = _UnpatchPad.apply(x, p, self.height // p, self.width // p, u, pad)
        ds = self.downsample
        extra_t, extra_s = more_pads if more_pads is not None else ([], [])
        kd, *extra = ops.pad_last2_group([ds.kernel] + list(extra_t), [(cu + pad, ds.kernel.shape[1])] + list(extra_s))   # one launch each way
        feat = feat.to(ds.dtype)
        k5 = kd.view(1, 1, 1, *kd.shape)
        if ops.conv3d_pointwise_fork_ok(feat, k5):
            # the features go on to the UNet through this node: the UNet's gradient joins the projection's inside one launch
            coarse, feat = ops.conv3d_pointwise_fork(feat, k5, ds.bias)
        else:
            coarse = ops.conv3d(feat, k5, ds.bias)
        return (feat, coarse) if more_pads is None else (feat, coarse, tuple(extra))

    def forward(self, x):
        x = self.upsample(self.linear(x))
        feat = rearrange(x, "b t (h w) (p1 p2 c u) -> b t (h p1) (w p2) (c u)", p1=self.patch_size, p2=self.patch_size,
                         h=self.height // self.patch_size, w=self.width // self.patch_size, u=self.upsample_rate)
        if feat.is_cuda:
            # Linear(c*u -> c) over every voxel of the full-resolution volume = a 1x1x1 conv: HBM-bound, far too skinny
            # for a BLAS GEMM (12 -> 3 channels over 4M rows), so it runs on the pointwise path of the conv kernels.
            ds = self.downsample
            coarse = ops.conv3d(feat.to(ds.dtype), ds.kernel.view(1, 1, 1, *ds.kernel.shape), ds.bias)
        else:
            coarse = self.downsample(feat)
        return feat, coarse


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class RotaryEmbedding(nn.Module):
    """Reference train/layers.py:85-129: NTK base, cos/sin caches (max_len, head_dim), rotate-half convention."""

    def __init__(self, head_dim, max_len=8192, alpha=1.0, base=10000.0):
        super().__init__()
        self.head_dim, self.max_len = head_dim, max_len
        ntk_base = base * (alpha ** (head_dim / (head_dim - 2)))
        inv_freq = 1.0 / (ntk_base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
        freqs = torch.einsum("i,j->ij", torch.arange(max_len, dtype=torch.float32), inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        self.register_buffer("cos_cached", torch.cos(emb).contiguous(), persistent=False)
        self.register_buffer("sin_cached", torch.sin(emb).contiguous(), persistent=False)

    def rotate_queries_and_keys(self, q, k):
        s = q.shape[1]
        cos = self.cos_cached[:s].to(q.dtype)[None, :, None, :]
        sin = self.sin_cached[:s].to(q.dtype)[None, :, None, :]
        return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


FUSED_CORE_MAX_SEQ = 64


class Attention(nn.Module):
    """Reference train/layers.py:131-171.  x (a, seq, dim); mask bool (a|b,1,1,seq) True = attend, or None."""

    def __init__(self, in_features, num_heads, qkv_features, max_len, use_qk_norm, rngs, dtype=torch.bfloat16,
                 param_dtype=torch.float32):
        super().__init__()
        self.num_heads = num_heads
        head_dim = qkv_features // num_heads
        self.qkv_projection = Linear(in_features, qkv_features * 3, rngs, dtype, param_dtype)
        self.qkv_projection.kernel.want_t = True   # (out, in) bf16 shadow: the library's K-contiguous kernel for the forward product
        self.out_projection = Linear(qkv_features, in_features, rngs, dtype, param_dtype, kernel_scale=1e-2)
        self.out_projection.kernel.want_t = True   # (out, in) shadow: the own NT GEMM's weight operand (residual epilogue)
        self.input_norm = LayerNorm(in_features, dtype, param_dtype)
        self.ROPE = RotaryEmbedding(head_dim=head_dim, max_len=max_len)
        self.use_qk_norm = use_qk_norm      # legacy flag in the reference; q/k norm is always applied
        self.q_norm = LayerNorm(head_dim, dtype, param_dtype, use_bias=False)
        self.k_norm = LayerNorm(head_dim, dtype, param_dtype, use_bias=False)

    def forward_temporal_strided(self, x, mask=None, pending=None, defer=False):
        """x + temporal attention on x laid out (b, t, hw, c) -- attention over t for every (b, hw) without the
        "b t hw c -> (b hw) t c" transposes: LayerNorm and the projections are per-token, the fused core strides.
        pending / defer: see LayerNorm.fork; defer=True returns the un-added (skip, branch) pair."""
        b, t, hw, _ = (pending[0] if pending is not None else x).shape
        x, skip = self.input_norm.fork(x, pending)
        qkv = self.qkv_projection(x)
        m8, div = None, 1
        if mask is not None:
            m8 = _mask_u8(mask, t)
            div = (b * hw) // m8.shape[0]
        o = ops.temporal_attention_core(qkv, self.q_norm.scale, self.k_norm.scale, self.ROPE.cos_cached, self.ROPE.sin_cached,
                                        m8, div, self.num_heads, 1e-6, inner=hw)
        return close_branch(self.out_projection, o, skip, defer)

    def residual(self, x, mask=None, pending=None, defer=False):
        """x + self(x) with the skip gradient folded into the input LayerNorm's backward (pending / defer: LayerNorm.fork)."""
        return self.forward(x, mask, _residual=True, _pending=pending, _defer=defer)

    def forward(self, x, mask=None, _residual=False, _pending=None, _defer=False):
        skip = None
        if _residual:
            x, skip = self.input_norm.fork(x, _pending)
        else:
            x = self.input_norm(x)
        qkv = self.qkv_projection(x)
        a, s, _ = qkv.shape
        if s <= FUSED_CORE_MAX_SEQ:
            # temporal half: fused HIP core (no fallback: raises off-GPU)
            m8, div = None, 1
            if mask is not None:
                m8 = mask.reshape(-1, s).to(torch.uint8).contiguous()
                div = a // m8.shape[0]
            o = ops.temporal_attention_core(qkv, self.q_norm.scale, self.k_norm.scale, self.ROPE.cos_cached,
                                            self.ROPE.sin_cached, m8, div, self.num_heads, 1e-6)
        elif mask is None and ops.spatial_attention_supported(qkv, self.num_heads, self.ROPE.cos_cached.shape[0]):
            # spatial half in bf16: the fused HIP kernel (q/k-norm + RoPE + softmax(QK^T)V; head_dim 64, S <= 256), or for other
            # shapes one HIP prep launch each way around the library flash-attention core (ops.spatial_attention_core decides)
            o = ops.spatial_attention_core(qkv, self.q_norm.scale, self.k_norm.scale, self.ROPE.cos_cached, self.ROPE.sin_cached,
                                           self.num_heads, 1e-6)
        else:
            if qkv.is_cuda and qkv.dtype == torch.bfloat16:
                ops.note_fallback(("sdpa", s, mask is not None), f"attention over sequence {s} ({'masked' if mask is not None else 'unmasked'}, bf16): "
                                  "neither fused HIP core takes it, the framework's scaled_dot_product_attention runs")
            q, k, v = torch.chunk(qkv, 3, dim=-1)
            q = rearrange(q, "b s (h d) -> b s h d", h=self.num_heads)
            k = rearrange(k, "b s (h d) -> b s h d", h=self.num_heads)
            v = rearrange(v, "b s (h d) -> b s h d", h=self.num_heads)
            q, k = self.ROPE.rotate_queries_and_keys(self.q_norm(q), self.k_norm(k))
            am = None
            if mask is not None:
                am = mask.to(torch.bool).expand(a, 1, 1, s) if mask.shape[0] == a else \
                    mask.to(torch.bool).repeat_interleave(a // mask.shape[0], dim=0)
            o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=am)
            o = rearrange(o, "b h s d -> b s (h d)")
        if not _residual:
            return self.out_projection(o)
        return close_branch(self.out_projection, o, skip, _defer)


class MLP(nn.Module):
    """Reference train/layers.py:174-196: LN -> Linear -> SiLU -> Linear (small-init)."""

    def __init__(self, in_features, mlp_dim, rngs, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.norm = LayerNorm(in_features, dtype, param_dtype)
        self.linear1 = Linear(in_features, mlp_dim, rngs, dtype, param_dtype)
        self.linear2 = Linear(mlp_dim, in_features, rngs, dtype, param_dtype, kernel_scale=1e-2)
        self.linear1.kernel.want_t = True          # the optimizer keeps an (out, in) bf16 shadow: fc1 + SiLU on the own NT GEMM
        self.linear2.kernel.want_t = True          # fc2 + residual: the library's faster operand form (ops.linear_residual wt)

    def forward(self, x):
        return silu_linear(self.linear1(self.norm(x)), self.linear2)

    def residual(self, x, pending=None, defer=False):
        """x + self(x) with the skip gradient folded into the LayerNorm's backward (pending / defer: LayerNorm.fork)."""
        y, skip = self.norm.fork(x, pending)
        h, a = linear_with_silu(self.linear1, y)
        return close_branch(self.linear2, h, skip, defer, silu=True, act=a)


class FactoredAttention(nn.Module):
    """Reference train/layers.py:198-224: temporal attn+MLP on (b*hw, t, c), then spatial attn+MLP on (b*t, hw, c).

    ``temporal_mask`` is (b*hw,1,1,t) as train_step builds it (rl_nonadversarial.py:190-192); the (b,1,1,t) form of
    claude_distributed/layers.py:213-214 is accepted too.
    """

    def __init__(self, mlp_dim, in_features, num_heads, qkv_features, max_temporal_len, max_spatial_len, rngs,
                 dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.SpatialAttention = Attention(in_features, num_heads, qkv_features, max_spatial_len, True, rngs, dtype, param_dtype)
        self.SpatialMLP = MLP(in_features, mlp_dim, rngs, dtype, param_dtype)
        self.TemporalAttention = Attention(in_features, num_heads, qkv_features, max_temporal_len, False, rngs, dtype, param_dtype)
        self.TemporalMLP = MLP(in_features, mlp_dim, rngs, dtype, param_dtype)

    def forward(self, x, temporal_mask, pending=None, defer=False):
        """pending = (skip, o): the input is the un-added residual pair the previous layer returned with defer=True (its last
        add then happens inside this layer's first LayerNorm kernel); defer=True returns such a pair instead of the sum."""
        first = pending[0] if pending is not None else x
        b, t, hw, c = first.shape
        ta = self.TemporalAttention
        hd = ta.q_norm.scale.shape[0]
        if first.is_cuda and t <= FUSED_CORE_MAX_SEQ and ops.temporal_attention_fast_supported(t, hd, 3 * hd * ta.num_heads, ta.qkv_projection.dtype):
            # every op of the temporal half except the attention core is per-token, and the core strides over frames:
            # stay in (b, t, hw, c) and skip both transpose copies (and their backward); each residual add rides in the
            # LayerNorm kernel of the following block
            p = ta.forward_temporal_strided(x, mask=temporal_mask, pending=pending, defer=True)
            p = self.TemporalMLP.residual(None, pending=p, defer=True)
            p = (p[0].reshape(b * t, hw, c), None if p[1] is None else p[1].reshape(b * t, hw, c))
            p = self.SpatialAttention.residual(None, pending=p, defer=True)
            p = self.SpatialMLP.residual(None, pending=p, defer=True)
            p = (p[0].view(b, t, hw, c), None if p[1] is None else p[1].view(b, t, hw, c))
            return p if defer else (p[0] if p[1] is None else p[0] + p[1])
        if pending is not None:
            x = pending[0] if pending[1] is None else pending[0] + pending[1]
        tx = rearrange(x, "b t hw c -> (b hw) t c")
        tx = self.TemporalAttention.residual(tx, mask=temporal_mask)
        tx = self.TemporalMLP.residual(tx)
        x = rearrange(tx, "(b hw) t c -> b t hw c", b=b, hw=hw)
        sx = rearrange(x, "b t hw c -> (b t) hw c")
        sx = self.SpatialAttention.residual(sx)
        sx = self.SpatialMLP.residual(sx)
        out = rearrange(sx, "(b t) hw c -> b t hw c", b=b, t=t)
        return (out, torch.zeros_like(out)) if defer else out


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, g):
        return g


def round_ste(logits):
    """round forward, identity backward (reference train/layers.py:226-236)."""
    return _RoundSTE.apply(logits)


class GumbelSigmoidSTE(nn.Module):
    """Reference train/layers.py:238-252."""

    def __init__(self, temperature=1.0):
        super().__init__()
        self.temperature = temperature

    def forward(self, logits, rngs, train=True):
        if train:
            eps = 1e-20
            u = rngs.draw("gumbel_u", "uniform", logits.shape, logits.device)
            # log(clip(u) / (1 - clip(u))) as ONE kernel (logit clamps to [eps, 1 - eps] first: the same formula, bit for bit), and no
            # division by a temperature of 1: on (b, t) elements every framework op is a ~5 us launch of the train step
            noise = torch.logit(u, eps=eps)
            y = logits + noise
            return round_ste(torch.sigmoid(y if self.temperature == 1.0 else y / self.temperature))
        return torch.round(torch.sigmoid(logits if self.temperature == 1.0 else logits / self.temperature))
