"""torch.autograd.Function wrappers over the C ABI (include/vvae_hip.h).

PyTorch is plumbing here: it owns device memory and streams; every op below
runs a hand-written HIP kernel from libvvae_hip.so on the current stream.
Tensors must live on a GPU -- there is no CPU path.
"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

from ._lib import lib, check, VvaeError

DT = {torch.float32: 0, torch.bfloat16: 1}

_NOTED = set()


def note_fallback(key, message):
    """Say ONCE per process (rank 0, stderr) that a launch left the path the benchmark measures: a conv that is too large for the bf16
    matrix-core kernels' 32-bit buffer offsets and runs the fp32 generic kernels instead, a layer shape the fused HIP kernels do not take and
    that runs a framework op.  Fallbacks are allowed; invisible performance cliffs are not."""
    if key in _NOTED:
        return
    _NOTED.add(key)
    import os
    import sys
    if os.environ.get("RANK", "0") == "0":
        print(f"[video_vae_amd] fallback: {message}", file=sys.stderr, flush=True)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    if not t.is_cuda:
        raise VvaeError("video_vae_amd ops need GPU tensors (no CPU fallback); got device " + str(t.device))
    if t.dtype not in DT:
        raise VvaeError(f"unsupported activation dtype {t.dtype} (float32 or bfloat16)")
    return DT[t.dtype]


def rows(t):
    """(tensor, row pitch) with the tensor usable as (rows, C) with a uniform pitch; copies only if it must."""
    if t.dim() < 2:
        raise VvaeError("need at least 2 dims")
    ok = t.stride(-1) == 1 and t.stride(-2) >= t.size(-1)
    if ok:
        for i in range(t.dim() - 2):
            if t.size(i) != 1 and t.stride(i) != t.stride(i + 1) * t.size(i + 1):
                ok = False
                break
    if not ok:
        t = t.contiguous()
    return t, t.stride(-2)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _ws(nbytes, device):
    if nbytes == 0:
        return None, 0
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf, nbytes


def fold_partials(part, p0, p1, n0):
    """Column sums of the per-workgroup partial rows ``part`` (rows, cols): columns [0, n0) are the gradient of parameter
    ``p0``, the rest that of ``p1`` (None: dropped).  Inside ``deferred_wgrad`` (and when the parameters own flat-buffer slots)
    the fold is parked and runs with up to 63 others in one launch straight into those slots: returns (None, None).  Otherwise
    returns the two sums."""
    q = WGRAD_QUEUE[0]
    if (q is not None and p0 is not None and part.is_cuda and getattr(p0, "gview", None) is not None and p0.numel() == n0
            and (p1 is None or (getattr(p1, "gview", None) is not None and p1.numel() == part.numel() // part.shape[0] - n0))):
        q.append_fold(part, p0, p1)
        return None, None
    tot = sum_rows(part).reshape(-1)
    return tot[:n0], tot[n0:]


def copy_grouped(dsts, srcs):
    """dst[i].copy_(src[i]) for lists of contiguous fp32 GPU tensors of equal sizes, 64 pairs per launch (gradient landing)."""
    for i0 in range(0, len(dsts), FOLD_MAX):
        d, s_ = dsts[i0:i0 + FOLD_MAX], srcs[i0:i0 + FOLD_MAX]
        n = len(d)
        VP, LA = ctypes.c_void_p * n, ctypes.c_long * n
        check(lib().vvae_copy_grouped(VP(*[t.data_ptr() for t in s_]), VP(*[t.data_ptr() for t in d]), LA(*[t.numel() for t in d]), n,
                                      _stream()), "vvae_copy_grouped")


def copy_grouped_ok(dst, src):
    return (dst.is_cuda and src.is_cuda and dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.is_contiguous()
            and src.is_contiguous() and dst.numel() == src.numel() and dst.numel() > 0)


def sum_rows(part):
    """Column sums of a (rows, ...) fp32 partial buffer in fixed order -> shape part.shape[1:] (no memset, no atomics).

    Tall-and-narrow buffers are folded in two launches: (rows, cols) is read as (rows/k, k*cols) first, so the first pass
    has enough workgroups, then the k partial rows are summed."""
    r = part.shape[0]
    cols0 = part.numel() // r
    cols, k = cols0, 1
    while r > 1024 and r % 2 == 0 and cols * 2 <= 8192:
        r //= 2; cols *= 2; k *= 2
    out = torch.empty((k,) + tuple(part.shape[1:]), dtype=torch.float32, device=part.device)
    check(lib().vvae_sum_rows(_p(part), r, cols, _p(out), _stream()), "vvae_sum_rows")
    if k == 1:
        return out[0]
    out2 = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    check(lib().vvae_sum_rows(_p(out), k, cols0, _p(out2), _stream()), "vvae_sum_rows")
    return out2


class KernelTimer:
    """Per-launch HIP-event timing of tagged kernels on the current stream (bench.py's roofline leg).

    ``records[tag] = [algorithmic_bytes, flops, kernel_name, [(start, end), ...]]``; durations are read after a sync.
    """

    def __init__(self, gate_us=25):
        self.records = {}
        # a short delay kernel in front of every timed launch: the opening event, the launch and the closing event are then all queued
        # before the first of them executes, and the interval is the kernel's duration -- not kernel + the host's launch latency, which is
        # what an event pair reads on an idle stream (eager steps are host-bound: the stream IS idle at every launch).  0 = off.
        self.gate_us = int(gate_us)

    def launch(self, tag, alg_bytes, flops, kernel, fn):
        rec = self.records.setdefault(tag, [alg_bytes, flops, kernel, []])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if self.gate_us > 0:
            check(lib().vvae_delay_us(self.gate_us, _stream()), "vvae_delay_us")
        a.record()
        rc = fn()
        b.record()
        rec[3].append((a, b))
        return rc

    def summary(self):
        """-> {tag: dict(n, avg_ms, total_ms, bytes, flops, kernel)} (call after torch.cuda.synchronize())."""
        out = {}
        for tag, (nbytes, flops, kernel, evs) in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[tag] = dict(n=len(ms), avg_ms=sum(ms) / len(ms), total_ms=sum(ms), bytes=nbytes, flops=flops, kernel=kernel)
        return out


TIMER = None      # set to a KernelTimer to time tagged launches


def _launch(tag, alg_bytes, flops, kernel, fn):
    if TIMER is None:
        return fn()
    return TIMER.launch(tag, alg_bytes, flops, kernel, fn)


BF16_SPAN_LIMIT = 1 << 31          # the bf16 conv kernels address through buffer descriptors: 32-bit byte offsets (conv3d_bf16.hip launch_roll)


def _bf16_fast(cin, cout, kt, kh, kw, ld_in, ld_out, which, dt, vox=0):
    """May this conv run the bf16 matrix-core kernels?  ``vox``: voxels of the launch -- a tensor of 2 GB or more (more than 512 frames of a
    32-channel 256 x 256 level) is beyond their 32-bit offsets and takes the generic fp32-matrix-core kernels, which is said once."""
    ok = dt == 1 and not _FORCE_GENERIC[0] and lib().vvae_conv3d_bf16_supported(cin, cout, kt, kh, kw, ld_in, ld_out, which, 0)
    if ok and vox and ((vox - 1) * max(ld_in, ld_out) + max(cin, cout)) * 2 >= BF16_SPAN_LIMIT:
        note_fallback(("conv3d>2GB", cin, cout, kt, kh, kw),
                      f"conv3d {cin}->{cout} k{kt}{kh}{kw} over {vox} voxels spans >= 2 GB: the bf16 MFMA kernels (32-bit buffer offsets) "
                      "decline it, the fp32 generic kernels run instead (several times slower) -- split the batch to stay on the fast path")
        return False
    return ok


_FORCE_GENERIC = [False]


def force_generic_conv(on):
    """Test hook: route every conv through the generic fp32-matrix-core path."""
    _FORCE_GENERIC[0] = bool(on)
    lib().vvae_conv3d_force_generic(1 if on else 0)


def _conv_fwd_like(x, ldx, kernel, bias, out, dims, dgrad, packed=None, k_real=0, price=None):
    """Shared by fwd (dgrad=0) and dgrad (dgrad=1): bf16 fast path (weights packed here, or already packed for the whole step by
    conv3d_prepack: ``packed``), else the dispatcher.  ``k_real``: how many of the layer's K channels are not zero padding (0 = all):
    the matrix-core kernels that know the count skip the padded part of the product (the 12-of-16 channel patch mixer).
    ``price``: (Cin, Cout) the layer really has when its tensors are zero-padded to the kernels' 16-channel granule: the algorithmic
    bytes / FLOPs bench.py quotes are those of the true layer, not of the padded launch."""
    n, t, h, w, cin, cout, kt, kh, kw = dims
    dt = _dt(x)
    ldo = out.stride(-2)
    esz = x.element_size()
    ck, co = (cout, cin) if dgrad else (cin, cout)
    vox = n * t * h * w
    pci, pco = price if price is not None else (cin, cout)
    alg = vox * (pci + pco) * esz
    flops = 2 * vox * kt * kh * kw * pci * pco
    name = "dgrad" if dgrad else "fwd"
    tag = f"conv3d_{name} {ck}->{co} k{kt}{kh}{kw} @{h}x{w}"
    if _bf16_fast(cin, cout, kt, kh, kw, ldx, ldo, 1 if dgrad else 0, dt, vox):
        flags = (1 if dgrad else 0) | (int(k_real) << 8)             # include/vvae_hip.h: vvae_conv3d_fwd_bf16
        if packed is not None:
            ws, wsb = packed, packed.numel()
        else:
            wsb = lib().vvae_conv3d_bf16_ws_bytes(n, t, h, w, cin, cout, kt, kh, kw, 1 if dgrad else 0)
            ws, wsb = _ws(wsb, x.device)
            check(lib().vvae_conv3d_pack_bf16(_p(kernel), _p(ws), wsb, cin, cout, kt, kh, kw, flags, _stream()),
                  "vvae_conv3d_pack_bf16")
        check(_launch(tag, alg, flops, "conv3d_bf16_roll_kernel|conv3d_bf16_deep_kernel",
                      lambda: lib().vvae_conv3d_fwd_bf16(_p(x), ldx, None, _p(bias), _p(out), ldo, n, t, h, w, cin, cout, kt, kh,
                                                         kw, flags, 1, _p(ws), wsb, _stream())),
              "vvae_conv3d_fwd_bf16")
        return out
    fn = lib().vvae_conv3d_dgrad if dgrad else lib().vvae_conv3d_fwd
    if dgrad:
        call = lambda: fn(_p(x), ldx, _p(kernel), _p(out), ldo, n, t, h, w, cin, cout, kt, kh, kw, dt, None, 0, _stream())
    else:
        call = lambda: fn(_p(x), ldx, _p(kernel), _p(bias), _p(out), ldo, n, t, h, w, cin, cout, kt, kh, kw, dt, None, 0, _stream())
    check(_launch(tag, alg, flops, "conv3d_f32mfma_kernel", call), "vvae_conv3d_" + name)
    return out


# --------------------------------------------------------------------------------------------- grouped zero-padding of small weights
def _pad_group_launch(srcs, dsts, rows_, s01, d01, unpad):
    n = len(srcs)
    VP, LA, IA = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    check(lib().vvae_pad_last2_grouped(VP(*[t.data_ptr() for t in srcs]), VP(*[t.data_ptr() for t in dsts]), LA(*rows_),
                                       IA(*[a for a, _ in s01]), IA(*[b for _, b in s01]), IA(*[a for a, _ in d01]), IA(*[b for _, b in d01]),
                                       n, 1 if unpad else 0, _stream()), "vvae_pad_last2_grouped")


def _last2(shape):
    return (shape[-2], shape[-1]) if len(shape) >= 2 else (1, shape[-1])


class _PadLast2Group(torch.autograd.Function):
    """Zero-pad the last two dims of up to 8 small fp32 parameters in one launch (vvae_pad_last2_grouped); the backward cuts the padded
    gradients back in one launch too -- inside ``deferred_wgrad`` straight into the parameters' slots of the flat gradient buffer."""

    @staticmethod
    def forward(ctx, targets, *srcs):
        ctx.params = srcs
        ctx.targets = targets
        s32 = [_f32(t) for t in srcs]
        outs = [torch.empty(tuple(t.shape[:-len(tg)]) + tuple(tg), dtype=torch.float32, device=t.device) for t, tg in zip(s32, targets)]
        s01 = [_last2(t.shape) for t in s32]
        d01 = [_last2(o.shape) for o in outs]
        _pad_group_launch(s32, outs, [t.numel() // (a * b) for t, (a, b) in zip(s32, s01)], s01, d01, False)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        q = WGRAD_QUEUE[0]
        todo, res = [], []
        for p, g in zip(ctx.params, gs):
            if g is None:
                res.append(None)
                continue
            g = g.to(torch.float32).contiguous()
            direct = q is not None and getattr(p, "gview", None) is not None and p.dtype == torch.float32
            dst = p.gview.view(p.shape) if direct else torch.empty(p.shape, dtype=torch.float32, device=g.device)
            todo.append((g, dst, p, direct))
            res.append(None if direct else dst)
        if todo:
            s01 = [_last2(d.shape) for _, d, _, _ in todo]
            d01 = [_last2(g.shape) for g, _, _, _ in todo]
            _pad_group_launch([g for g, _, _, _ in todo], [d for _, d, _, _ in todo],
                              [d.numel() // (a * b) for (_, d, _, _), (a, b) in zip(todo, s01)], s01, d01, True)
            for _, _, p, direct in todo:
                if direct:
                    q.opt.mark_external(p)
        return (None,) + tuple(r if r is None else r.to(p.dtype) for r, p in zip(res, ctx.params))


def pad_last2_group(tensors, targets):
    """-> zero-padded fp32 copies of ``tensors`` (GPU parameters): ``targets[i]`` is the new size of the last two dims (one dim for a
    vector).  One launch forward, one backward, however many tensors (<= 8)."""
    return _PadLast2Group.apply(tuple(tuple(t) for t in targets), *tensors)


# --------------------------------------------------------------------------------------------- Conv3d
def conv3d_fwd_raw(x, kernel, bias, out=None, packed=None, k_real=0, price=None):
    x, ldx = rows(x)
    n, t, h, w, cin = x.shape
    kt, kh, kw, cin2, cout = kernel.shape
    assert cin == cin2, (cin, cin2)
    if out is None:
        out = torch.empty((n, t, h, w, cout), dtype=x.dtype, device=x.device)
    return _conv_fwd_like(x, ldx, kernel, bias, out, (n, t, h, w, cin, cout, kt, kh, kw), 0, packed, k_real, price)


def conv3d_dgrad_raw(dy, kernel, out=None, packed=None, k_real=0, price=None):
    dy, lddy = rows(dy)
    n, t, h, w, cout = dy.shape
    kt, kh, kw, cin, cout2 = kernel.shape
    assert cout == cout2
    if out is None:
        out = torch.empty((n, t, h, w, cin), dtype=dy.dtype, device=dy.device)
    return _conv_fwd_like(dy, lddy, kernel, None, out, (n, t, h, w, cin, cout, kt, kh, kw), 1, packed, k_real, price)


class ConvPack:
    """Packed bf16 weight fragments of one conv layer for one optimizer step: ``fwd`` and ``dgrad`` are uint8 views of one buffer."""
    __slots__ = ("fwd", "dgrad")

    def __init__(self, fwd, dgrad):
        self.fwd, self.dgrad = fwd, dgrad


def conv3d_prepack(kernels, reals=None):
    """Pack the weights of every conv layer of a network -- forward and flipped input-gradient forms -- in ONE launch
    (vvae_conv3d_pack_grouped_bf16).  ``kernels``: fp32 GPU tensors (3, kh, kw, Cin, Cout); -> list of ConvPack (None for a layer
    the bf16 matrix-core kernels do not take).  Weights change once per optimizer step; packing per call was 28 launches a step.
    ``reals``: per kernel None or (real Cin, real Cout) of a zero-padded layer (conv3d's ``real``)."""
    reals = reals if reals is not None else [None] * len(kernels)
    specs, sizes = [], []
    for k in kernels:
        kt, kh, kw, cin, cout = k.shape
        ok = (k.is_cuda and k.dtype == torch.float32 and kt == 3 and kh == kw and kh in (3, 7) and not _FORCE_GENERIC[0]
              and lib().vvae_conv3d_bf16_supported(cin, cout, kt, kh, kw, 8, 8, 0, 0) == 1
              and lib().vvae_conv3d_bf16_supported(cin, cout, kt, kh, kw, 8, 8, 1, 0) == 1)
        specs.append(ok)
        if ok:
            for which in (0, 1):
                sizes.append((lib().vvae_conv3d_bf16_ws_bytes(1, 1, 1, 1, cin, cout, kt, kh, kw, which) + 255) // 256 * 256)
    if not sizes:
        return [None] * len(kernels)
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=kernels[0].device)
    packs, off, ent = [], 0, []
    it = iter(sizes)
    for k, ok, real in zip(kernels, specs, reals):
        if not ok:
            packs.append(None)
            continue
        views = []
        for which in (0, 1):
            nb = next(it)
            v = buf[off:off + nb]
            off += nb
            views.append(v)
            ent.append((k.contiguous(), v, which | ((int(real[which]) << 8) if real is not None else 0)))
        packs.append(ConvPack(views[0], views[1]))
    for i0 in range(0, len(ent), 64):
        e = ent[i0:i0 + 64]
        n = len(e)
        VP, IA, SA = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_size_t * n
        check(lib().vvae_conv3d_pack_grouped_bf16(VP(*[k.data_ptr() for k, _, _ in e]), VP(*[v.data_ptr() for _, v, _ in e]),
                                                  SA(*[v.numel() for _, v, _ in e]), IA(*[k.shape[3] for k, _, _ in e]),
                                                  IA(*[k.shape[4] for k, _, _ in e]), IA(*[k.shape[1] for k, _, _ in e]),
                                                  IA(*[w for _, _, w in e]), n, _stream()), "vvae_conv3d_pack_grouped_bf16")
    return packs


def conv3d_wgrad_raw(x, dy, kshape, want_bias=True, dw_out=None, db_out=None, price=None):
    """-> (dw, db) fp32; dw_out / db_out: contiguous fp32 buffers to overwrite instead of fresh ones (flat-buffer slots)."""
    x, ldx = rows(x)
    dy, lddy = rows(dy)
    n, t, h, w, cin = x.shape
    kt, kh, kw, _, cout = kshape
    dw = dw_out if dw_out is not None else torch.empty(kshape, dtype=torch.float32, device=x.device)
    db = (db_out if db_out is not None else torch.empty((cout,), dtype=torch.float32, device=x.device)) if want_bias else None
    dt = _dt(x)
    wsb = lib().vvae_conv3d_workspace_bytes(n, t, h, w, cin, cout, kt, kh, kw, dt, 2)
    ws, wsb = _ws(wsb, x.device)
    vox = n * t * h * w
    _bf16_fast(cin, cout, kt, kh, kw, ldx, lddy, 2, dt, vox)        # says so once if the launch is too large for the bf16 kernels (the dispatcher then goes generic)
    tag = f"conv3d_wgrad {cin}->{cout} k{kt}{kh}{kw} @{h}x{w}"
    pci, pco = price if price is not None else (cin, cout)
    check(_launch(tag, vox * (pci + pco) * x.element_size(), 2 * vox * kt * kh * kw * pci * pco, "conv3d_wgrad",
                  lambda: lib().vvae_conv3d_wgrad(_p(x), ldx, _p(dy), lddy, _p(dw), _p(db), n, t, h, w, cin, cout, kt, kh, kw, dt,
                                                  _p(ws), wsb, _stream())), "vvae_conv3d_wgrad")
    return dw, db


def conv3d_gn_blocks(x, kernel, groups):
    """Rows per sample of the GroupNorm partial buffer the rolling forward kernel can emit for this layer; 0: not eligible."""
    if not (x.is_cuda and x.dtype == torch.bfloat16) or _FORCE_GENERIC[0] or groups <= 0:
        return 0
    n, t, h, w, cin = x.shape
    kt, kh, kw, _, cout = kernel.shape
    return lib().vvae_conv3d_gn_blocks(n, t, h, w, cin, cout, kt, kh, kw, x.stride(-2), cout, groups)


def conv3d_fwd_gn_raw(x, kernel, bias, groups, nblk, packed=None, price=None):
    """conv3d_fwd_raw + the per-(sample, workgroup, group) sums of the rounded outputs: -> (y, part (n, nblk, groups, 2) fp32)."""
    x, ldx = rows(x)
    n, t, h, w, cin = x.shape
    kt, kh, kw, _, cout = kernel.shape
    out = torch.empty((n, t, h, w, cout), dtype=x.dtype, device=x.device)
    part = torch.empty((n, nblk, groups, 2), dtype=torch.float32, device=x.device)
    if packed is not None:
        ws, wsb = packed, packed.numel()
    else:
        wsb = lib().vvae_conv3d_bf16_ws_bytes(n, t, h, w, cin, cout, kt, kh, kw, 0)
        ws, wsb = _ws(wsb, x.device)
    vox = n * t * h * w
    tag = f"conv3d_fwd {cin}->{cout} k{kt}{kh}{kw} @{h}x{w}"
    pci, pco = price if price is not None else (cin, cout)
    check(_launch(tag, vox * (pci + pco) * 2, 2 * vox * kt * kh * kw * pci * pco, "conv3d_bf16_roll_kernel|conv3d_bf16_deep_kernel",
                  lambda: lib().vvae_conv3d_fwd_bf16_gn(_p(x), ldx, _p(kernel), _p(bias), _p(out), cout, n, t, h, w, cin, cout, kt, kh,
                                                        kw, 1 if packed is not None else 0, _p(ws), wsb, _p(part), groups, _stream())),
          "vvae_conv3d_fwd_bf16_gn")
    return out, part


class _Conv3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, gn_groups=0, gn_blocks=0, pack=None, real=None, price=None):
        k32 = _f32(kernel)
        b32 = _f32(bias) if bias is not None else None
        ctx.save_for_backward(x, k32)
        ctx.real = real                                  # (real Cin, real Cout) of a zero-padded layer, or None
        ctx.price = price if price is not None else real    # (Cin, Cout) the roofline accounting uses (_conv_fwd_like)
        ctx.has_bias = bias is not None
        ctx.kdtype = kernel.dtype
        ctx.kparam, ctx.bparam = kernel, bias            # for ops.deferred_wgrad: where the gradient may be written directly
        ctx.pack = pack                                  # this step's packed weights (conv3d_prepack), or None: pack per call
        if gn_blocks:
            y, part = conv3d_fwd_gn_raw(x, k32, b32, gn_groups, gn_blocks, pack.fwd if pack is not None else None, ctx.price)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)             # no zero-filled gradient tensor for the partials in backward
            ctx.with_part = True
            return y, part
        ctx.with_part = False
        return conv3d_fwd_raw(x, k32, b32, packed=pack.fwd if pack is not None else None, k_real=real[0] if real is not None else 0,
                              price=ctx.price)

    @staticmethod
    def backward(ctx, dy, dpart=None):
        dx, dw, db = _Conv3d._backward(ctx, dy)
        return dx, dw, db, None, None, None, None, None

    @staticmethod
    def _backward(ctx, dy):
        x, k32 = ctx.saved_tensors
        dy = dy.to(x.dtype)
        dx = (conv3d_dgrad_raw(dy, k32, packed=ctx.pack.dgrad if ctx.pack is not None else None,
                               k_real=ctx.real[1] if ctx.real is not None else 0, price=ctx.price) if ctx.needs_input_grad[0] else None)
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            q = WGRAD_QUEUE[0]
            kp, bp = ctx.kparam, ctx.bparam
            kt, kh, kw, cin, cout = k32.shape
            if (q is not None and ctx.needs_input_grad[1] and getattr(kp, "gview", None) is not None and kp.dtype == torch.float32
                    and (bp is None or (ctx.needs_input_grad[2] and getattr(bp, "gview", None) is not None and bp.dtype == torch.float32))
                    and _bf16_fast(cin, cout, kt, kh, kw, x.stride(-2), dy.stride(-2), 2, _dt(x), x.numel() // cin)):
                # inside ops.deferred_wgrad the slab-reduce kernel overwrites the parameters' slots of the flat gradient buffer
                # directly: no gradient tensor, no landing copy (and no clone by autograd.grad inside a captured graph)
                q.claim(kp)
                conv3d_wgrad_raw(x, dy, tuple(k32.shape), bp is not None, kp.gview, bp.gview if bp is not None else None, ctx.price)
                q.opt.mark_external(kp)
                if bp is not None:
                    q.opt.mark_external(bp)
                return dx, None, None
            dw, db = conv3d_wgrad_raw(x, dy, tuple(k32.shape), ctx.has_bias, price=ctx.price)
            dw = dw.to(ctx.kdtype)
        return dx, dw, db


class _PointwiseAdd(torch.autograd.Function):
    """addend + conv1x1x1(x) in one launch (vvae_conv_pointwise_fwd_add): the decoder's ``coarse + UNet(features)`` (reference
    train/model.py:97) inside the product that ends the UNet.  Backward = _Conv3d's, and the gradient itself for the addend."""

    @staticmethod
    def forward(ctx, x, kernel, bias, addend):
        k32 = _f32(kernel)
        b32 = _f32(bias) if bias is not None else None
        ctx.save_for_backward(x, k32)
        ctx.real = ctx.price = ctx.pack = None
        ctx.has_bias = bias is not None
        ctx.kdtype = kernel.dtype
        ctx.kparam, ctx.bparam = kernel, bias
        xr, ldx = rows(x)
        ar, lda = rows(addend)
        cin, cout = k32.shape[-2], k32.shape[-1]
        out = torch.empty(addend.shape, dtype=x.dtype, device=x.device)
        vox = x.numel() // cin
        h, w = x.shape[-3], x.shape[-2]
        check(_launch(f"conv3d_fwd+add {cin}->{cout} k111 @{h}x{w}", vox * (cin + 2 * cout) * x.element_size(), 2 * vox * cin * cout, "pw_fwd_kernel",
                      lambda: lib().vvae_conv_pointwise_fwd_add(_p(xr), ldx, _p(k32), _p(b32), _p(ar), lda, _p(out), out.stride(-2), vox, cin, cout,
                                                                _dt(x), _stream())), "vvae_conv_pointwise_fwd_add")
        return out

    @staticmethod
    def backward(ctx, dy):
        dx, dw, db = _Conv3d._backward(ctx, dy)
        return dx, dw, db, dy


class _PointwiseFork(torch.autograd.Function):
    """(conv1x1x1(x), x) from one node, for a tensor with a second consumer (the un-patchified features feed the coarse projection AND the
    UNet, reference train/layers.py:60-79, train/model.py:95-97): the other consumer's gradient reaches THIS backward and is added inside
    the input-gradient launch (vvae_conv_pointwise_dgrad_add) instead of by the engine's 134 MB accumulation launch."""

    @staticmethod
    def forward(ctx, x, kernel, bias):
        k32 = _f32(kernel)
        b32 = _f32(bias) if bias is not None else None
        ctx.save_for_backward(x, k32)
        ctx.has_bias = bias is not None
        ctx.kdtype = kernel.dtype
        ctx.set_materialize_grads(False)
        return conv3d_fwd_raw(x, k32, b32), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dother):
        x, k32 = ctx.saved_tensors
        if dy is None:
            return dother, None, None
        dy = dy.to(x.dtype)
        xr, ldx = rows(x)
        dyr, lddy = rows(dy)
        cin, cout = k32.shape[-2], k32.shape[-1]
        vox = xr.numel() // cin
        dx = torch.empty(xr.shape, dtype=xr.dtype, device=xr.device)
        ad, lda = (None, 0) if dother is None else rows(dother.to(x.dtype))
        check(lib().vvae_conv_pointwise_dgrad_add(_p(dyr), lddy, _p(k32), _p(ad), lda, _p(dx), dx.stride(-2), vox, cin, cout, _dt(xr), _stream()),
              "vvae_conv_pointwise_dgrad_add")
        dw, db = conv3d_wgrad_raw(x, dy, tuple(k32.shape), ctx.has_bias)
        return dx, dw.to(ctx.kdtype), db


def conv3d_pointwise_fork_ok(x, kernel):
    kt, kh, kw, cin, cout = kernel.shape
    return (x.is_cuda and not _FORCE_GENERIC[0] and x.dtype in DT and x.shape[-1] == cin and x.is_contiguous()
            and lib().vvae_conv_pointwise_supported(cin, cout, kt, kh, kw, x.stride(-2), DT[x.dtype], _p(x)) == 1)


def conv3d_pointwise_fork(x, kernel, bias):
    """-> (conv3d(x, kernel (1,1,1,Cin,Cout), bias), x): use the returned x for the tensor's other consumer, whose gradient then joins inside
    the projection's input-gradient launch."""
    return _PointwiseFork.apply(x, kernel, bias)


def conv3d_pointwise_add_ok(x, kernel, addend):
    kt, kh, kw, cin, cout = kernel.shape
    return (x.is_cuda and not _FORCE_GENERIC[0] and x.dtype == addend.dtype and x.dtype in DT and x.shape[-1] == cin
            and addend.shape == (*x.shape[:-1], cout) and x.stride(-1) == 1 and addend.stride(-1) == 1
            and lib().vvae_conv_pointwise_supported(cin, cout, kt, kh, kw, x.stride(-2), DT[x.dtype], _p(x)) == 1)


def conv3d_pointwise_add(x, kernel, bias, addend):
    """addend + conv3d(x, kernel (1,1,1,Cin,Cout), bias), one pass and one rounding."""
    return _PointwiseAdd.apply(x, kernel, bias, addend)


# ---- conv over concat([xa, xb], channels) held as two dense tensors (the decoder's 16 + 16 channel level) ----
def conv3d_cat2_ok(xa, xb, kernel):
    """The rolling bf16 kernels take this layer with its input in two tensors (forward, input gradient and weight gradient)."""
    if _FORCE_GENERIC[0] or not (xa.is_cuda and xa.dtype == torch.bfloat16 and xb.dtype == torch.bfloat16 and xa.shape[:-1] == xb.shape[:-1]):
        return False
    kt, kh, kw, cin, cout = kernel.shape
    return cin == xa.shape[-1] + xb.shape[-1] and lib().vvae_conv3d_cat2_supported(cin, cout, xa.shape[-1], kt, kh, kw) == 1


def _cat2_pack(kernel, dgrad, dims, device):
    n, t, h, w, cin, cout, kt, kh, kw = dims
    wsb = lib().vvae_conv3d_bf16_ws_bytes(n, t, h, w, cin, cout, kt, kh, kw, dgrad)
    ws, wsb = _ws(wsb, device)
    check(lib().vvae_conv3d_pack_bf16(_p(kernel), _p(ws), wsb, cin, cout, kt, kh, kw, dgrad, _stream()), "vvae_conv3d_pack_bf16")
    return ws


def conv3d_cat2_fwd_raw(xa, xb, kernel, bias, groups=0, nblk=0, packed=None):
    """-> y, or (y, part) with the GroupNorm partials when nblk > 0 (conv3d_fwd_gn_raw)."""
    xa, lda = rows(xa)
    xb, ldb = rows(xb)
    n, t, h, w, ca = xa.shape
    kt, kh, kw, cin, cout = kernel.shape
    dims = (n, t, h, w, cin, cout, kt, kh, kw)
    ws = packed if packed is not None else _cat2_pack(kernel, 0, dims, xa.device)
    out = torch.empty((n, t, h, w, cout), dtype=xa.dtype, device=xa.device)
    part = torch.empty((n, nblk, groups, 2), dtype=torch.float32, device=xa.device) if nblk else None
    vox = n * t * h * w
    tag = f"conv3d_fwd {cin}->{cout} k{kt}{kh}{kw} @{h}x{w}"
    check(_launch(tag, vox * (cin + cout) * 2, 2 * vox * kt * kh * kw * cin * cout, "conv3d_bf16_roll_kernel|conv3d_bf16_deep_kernel",
                  lambda: lib().vvae_conv3d_fwd_bf16_cat2(_p(xa), lda, _p(xb), ldb, _p(bias), _p(out), cout, None, 0, ca, n, t, h, w, cin, cout,
                                                          kt, kh, kw, 0, _p(ws), ws.numel(), _p(part), groups, _stream())),
          "vvae_conv3d_fwd_bf16_cat2")
    return (out, part) if nblk else out


def conv3d_cat2_dgrad_raw(dy, kernel, ca, packed=None):
    """-> (dxa, dxb): the input gradient's first ``ca`` channels and the rest, each dense."""
    dy, lddy = rows(dy)
    n, t, h, w, cout = dy.shape
    kt, kh, kw, cin, _ = kernel.shape
    dims = (n, t, h, w, cin, cout, kt, kh, kw)
    ws = packed if packed is not None else _cat2_pack(kernel, 1, dims, dy.device)
    dxa = torch.empty((n, t, h, w, ca), dtype=dy.dtype, device=dy.device)
    dxb = torch.empty((n, t, h, w, cin - ca), dtype=dy.dtype, device=dy.device)
    vox = n * t * h * w
    tag = f"conv3d_dgrad {cout}->{cin} k{kt}{kh}{kw} @{h}x{w}"
    check(_launch(tag, vox * (cin + cout) * 2, 2 * vox * kt * kh * kw * cin * cout, "conv3d_bf16_roll_kernel|conv3d_bf16_deep_kernel",
                  lambda: lib().vvae_conv3d_fwd_bf16_cat2(_p(dy), lddy, None, 0, None, _p(dxa), ca, _p(dxb), cin - ca, ca, n, t, h, w, cin, cout,
                                                          kt, kh, kw, 1, _p(ws), ws.numel(), None, 0, _stream())),
          "vvae_conv3d_fwd_bf16_cat2")
    return dxa, dxb


def conv3d_cat2_wgrad_raw(xa, xb, dy, kshape, want_bias=True, dw_out=None, db_out=None):
    xa, lda = rows(xa)
    xb, ldb = rows(xb)
    dy, lddy = rows(dy)
    n, t, h, w, ca = xa.shape
    kt, kh, kw, cin, cout = kshape
    dw = dw_out if dw_out is not None else torch.empty(kshape, dtype=torch.float32, device=xa.device)
    db = (db_out if db_out is not None else torch.empty((cout,), dtype=torch.float32, device=xa.device)) if want_bias else None
    wsb = lib().vvae_conv3d_wgrad_bf16_ws_bytes(n, t, h, w, cin, cout, kt, kh, kw)
    ws, wsb = _ws(wsb, xa.device)
    vox = n * t * h * w
    tag = f"conv3d_wgrad {cin}->{cout} k{kt}{kh}{kw} @{h}x{w}"
    check(_launch(tag, vox * (cin + cout) * 2, 2 * vox * kt * kh * kw * cin * cout, "conv3d_wgrad",
                  lambda: lib().vvae_conv3d_wgrad_bf16_cat2(_p(xa), lda, _p(xb), ldb, ca, _p(dy), lddy, _p(dw), _p(db), n, t, h, w, cin, cout, kt,
                                                            kh, kw, _p(ws), wsb, _stream())), "vvae_conv3d_wgrad_bf16_cat2")
    return dw, db


class _Conv3dCat2(torch.autograd.Function):
    """_Conv3d over concat([xa, xb], -1) without the concatenated tensor (reference train/unet.py:79-81)."""

    @staticmethod
    def forward(ctx, xa, xb, kernel, bias, gn_groups=0, gn_blocks=0, pack=None):
        k32, b32 = _f32(kernel), _f32(bias)
        ctx.save_for_backward(xa, xb, k32)
        ctx.kdtype, ctx.kparam, ctx.bparam, ctx.pack = kernel.dtype, kernel, bias, pack
        pk = pack.fwd if pack is not None else None
        if gn_blocks:
            y, part = conv3d_cat2_fwd_raw(xa, xb, k32, b32, gn_groups, gn_blocks, pk)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y, part
        return conv3d_cat2_fwd_raw(xa, xb, k32, b32, packed=pk)

    @staticmethod
    def backward(ctx, dy, dpart=None):
        xa, xb, k32 = ctx.saved_tensors
        dy = dy.to(xa.dtype)
        dxa = dxb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dxa, dxb = conv3d_cat2_dgrad_raw(dy, k32, xa.shape[-1], ctx.pack.dgrad if ctx.pack is not None else None)
        dw = db = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            q = WGRAD_QUEUE[0]
            kp, bp = ctx.kparam, ctx.bparam
            if (q is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and getattr(kp, "gview", None) is not None
                    and getattr(bp, "gview", None) is not None and kp.dtype == torch.float32 and bp.dtype == torch.float32):
                q.claim(kp)                                  # straight into the flat gradient buffer (see _Conv3d._backward)
                conv3d_cat2_wgrad_raw(xa, xb, dy, tuple(k32.shape), True, kp.gview, bp.gview)
                q.opt.mark_external(kp)
                q.opt.mark_external(bp)
            else:
                dw, db = conv3d_cat2_wgrad_raw(xa, xb, dy, tuple(k32.shape), True)
                dw = dw.to(ctx.kdtype)
        return dxa, dxb, dw, db, None, None, None


def conv3d_cat2_with_gn_stats(xa, xb, kernel, bias, groups, pack=None):
    """conv3d_with_gn_stats over concat([xa, xb], -1) held as two tensors; the caller checked conv3d_cat2_ok."""
    kt, kh, kw, cin, cout = kernel.shape
    n, t, h, w, _ = xa.shape
    nblk = 0 if _FORCE_GENERIC[0] or groups <= 0 else lib().vvae_conv3d_gn_blocks(n, t, h, w, cin, cout, kt, kh, kw, 8, cout, groups)
    if not nblk:
        return _Conv3dCat2.apply(xa, xb, kernel, bias, 0, 0, pack), None
    y, part = _Conv3dCat2.apply(xa, xb, kernel, bias, groups, nblk, pack)
    return y, (part, nblk)


def conv3d(x, kernel, bias=None, pack=None, real=None):
    """NDHWC Conv3d, SAME, stride 1 (nnx.Conv; reference train/unet.py:13-21).  ``pack``: from conv3d_prepack.  ``real``: (Cin, Cout)
    actually in use when x / kernel / bias are zero-padded to the kernels' channel granule (must match the prepack call's)."""
    return _Conv3d.apply(x, kernel, bias, 0, 0, pack, real)


def conv3d_with_gn_stats(x, kernel, bias, groups, pack=None, price=None):
    """-> (conv3d(x), stats) where stats is None or (partial sums, rows per sample) for group_norm_silu(..., stats=...): on
    the rolling bf16 kernel the conv's epilogue also sums its rounded outputs per GroupNorm group, so the norm behind it
    (reference train/unet.py:13-23) skips its own statistics pass over the tensor."""
    nblk = conv3d_gn_blocks(x, kernel, groups) if bias is not None else 0
    if not nblk:
        return _Conv3d.apply(x, kernel, bias, 0, 0, pack, None, price), None
    y, part = _Conv3d.apply(x, kernel, bias, groups, nblk, pack, None, price)
    return y, (part, nblk)


def unpatch_pad(src, dst, frames, h, w, p, cu, c, backward):
    """PatchUnEmbedding's rearrange fused with the channel padding (layers._UnpatchPad); bf16 GPU tensors, contiguous."""
    fn = lib().vvae_unpatch_pad_bwd if backward else lib().vvae_unpatch_pad_fwd
    check(fn(_p(src), _p(dst), frames, h, w, p, cu, c, _dt(src), _stream()), "vvae_unpatch_pad")


# --------------------------------------------------------------------------------------------- GroupNorm + SiLU
def gn_stats_raw(x, groups):
    x, ldx = rows(x)
    n, c = x.shape[0], x.shape[-1]
    s = x.numel() // (n * c)
    sums = torch.empty((n, groups, 2), dtype=torch.float64, device=x.device)
    part = torch.empty((lib().vvae_gn_part_floats(n, s, c),), dtype=torch.float32, device=x.device)
    check(lib().vvae_gn_stats(_p(x), ldx, n, s, c, groups, _p(sums), _p(part), _dt(x), _stream()), "vvae_gn_stats")
    return sums


def gn_silu_fwd_raw(x, sums, scale, bias, groups, eps, out=None):
    x, ldx = rows(x)
    n, c = x.shape[0], x.shape[-1]
    s = x.numel() // (n * c)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(lib().vvae_gn_silu_fwd(_p(x), ldx, _p(out), out.stride(-2), _p(sums), _p(scale), _p(bias), n, s, c, groups, eps,
                                 _dt(x), _stream()), "vvae_gn_silu_fwd")
    return out


def gn_silu_bwd_raw(x, dy, sums, scale, bias, groups, eps, out=None):
    x, ldx = rows(x)
    dy, lddy = rows(dy)
    n, c = x.shape[0], x.shape[-1]
    s = x.numel() // (n * c)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    csum = torch.empty((n, c, 2), dtype=torch.float64, device=x.device)
    part = torch.empty((lib().vvae_gn_part_floats(n, s, c),), dtype=torch.float32, device=x.device)
    dg = torch.empty((c,), dtype=torch.float32, device=x.device)
    db = torch.empty((c,), dtype=torch.float32, device=x.device)
    check(lib().vvae_gn_silu_bwd(_p(x), ldx, _p(dy), lddy, _p(out), out.stride(-2), _p(sums), _p(scale), _p(bias), _p(csum),
                                 _p(part), _p(dg), _p(db), n, s, c, groups, eps, _dt(x), _stream()), "vvae_gn_silu_bwd")
    return out, dg, db


GN_POOL_FWD_FUSED = [True]       # test / A-B hook: False = GroupNorm + SiLU, then the pool as a launch of its own


def gn_silu_pool_ok(x, groups, out=None):
    """May group_norm_silu(..., pool=True) run?  (vvae_gn_silu_pool_fwd: 5-D NDHWC on the GPU, even H and W, 16-byte channel vectors.)"""
    if not (x.is_cuda and x.dim() == 5 and x.dtype in DT and x.stride(-1) == 1) or _FORCE_GENERIC[0] or not GN_POOL_FWD_FUSED[0]:
        return False
    h, w, c = x.shape[-3], x.shape[-2], x.shape[-1]
    ldy = out.stride(-2) if out is not None else c
    return lib().vvae_gn_silu_pool_supported(h, w, c, groups, x.stride(-2), ldy, c, DT[x.dtype]) == 1


class _GnSilu(torch.autograd.Function):
    """silu(GroupNorm(x)).  ``pool``: also return its (1,2,2) max-pool from the same launch (the encoder levels' conv2 -> GN -> SiLU ->
    max_pool, reference train/unet.py:44-51); the backward then first routes the pool's gradient to the window maxima and adds the skip's
    (maxpool_bwd_raw, as ops.max_pool_fork did) and runs the GroupNorm backward on the sum."""

    @staticmethod
    def forward(ctx, x, scale, bias, groups, eps, out=None, part=None, nblk=0, pool=False):
        s32, b32 = _f32(scale), _f32(bias)
        if part is not None:                             # the producing conv already summed its outputs (conv3d_with_gn_stats)
            sums = torch.empty((x.shape[0], groups, 2), dtype=torch.float64, device=x.device)
            check(lib().vvae_gn_finalize(_p(part), x.shape[0], nblk, groups, _p(sums), _stream()), "vvae_gn_finalize")
        else:
            sums = gn_stats_raw(x, groups)
        ctx.groups, ctx.eps, ctx.pdtype, ctx.pool = groups, eps, scale.dtype, bool(pool)
        ctx.set_materialize_grads(False)
        if not pool:
            ctx.save_for_backward(x, sums, s32, b32)
            return gn_silu_fwd_raw(x, sums, s32, b32, groups, eps, out)
        xr, ldx = rows(x)
        n, t, h, w, c = xr.shape
        y = out if out is not None else torch.empty(xr.shape, dtype=xr.dtype, device=xr.device)
        pooled = torch.empty((n, t, h // 2, w // 2, c), dtype=xr.dtype, device=xr.device)
        check(lib().vvae_gn_silu_pool_fwd(_p(xr), ldx, _p(y), y.stride(-2), _p(pooled), c, _p(sums), _p(s32), _p(b32), n, t, h, w, c, groups,
                                          eps, _dt(xr), _stream()), "vvae_gn_silu_pool_fwd")
        ctx.save_for_backward(x, sums, s32, b32, y)
        return y, pooled

    @staticmethod
    def backward(ctx, dy, dpool=None):
        if ctx.pool:
            x, sums, s32, b32, y = ctx.saved_tensors
            if dpool is not None:
                dy = maxpool_bwd_raw(y, dpool.to(y.dtype), None if dy is None else dy.to(y.dtype))
        else:
            x, sums, s32, b32 = ctx.saved_tensors
        if dy is None:
            return (None,) * 9
        dx, dg, db = gn_silu_bwd_raw(x, dy.to(x.dtype), sums, s32, b32, ctx.groups, ctx.eps)
        return dx, dg.to(ctx.pdtype), db.to(ctx.pdtype), None, None, None, None, None, None


def group_norm_silu(x, scale, bias, groups, eps=1e-6, out=None, stats=None, pool=False):
    """silu(GroupNorm(x)) over (t,h,w,C/G) per sample (reference train/unet.py:22-23,28-29).  ``out``: a channel slice of a
    wider NDHWC buffer to write into (see join_channels).  ``stats``: from conv3d_with_gn_stats.  ``pool`` (gn_silu_pool_ok): -> (y, max_pool_1x2x2(y))."""
    if stats is not None:
        return _GnSilu.apply(x, scale, bias, groups, eps, out, stats[0], stats[1], pool)
    return _GnSilu.apply(x, scale, bias, groups, eps, out, None, 0, pool)


# --------------------------------------------------------------------------------------------- max-pool (1,2,2)
def maxpool_fwd_raw(x, out=None):
    x, ldx = rows(x)
    n, t, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, t, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    check(lib().vvae_maxpool_1x2x2_fwd(_p(x), ldx, _p(out), out.stride(-2), n * t, h, w, c, _dt(x), _stream()),
          "vvae_maxpool_1x2x2_fwd")
    return out


def maxpool_bwd_raw(x, dpool, dskip=None, out=None):
    x, ldx = rows(x)
    dpool, lddp = rows(dpool)
    ldds = 0
    if dskip is not None:
        dskip, ldds = rows(dskip)
    n, t, h, w, c = x.shape
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(lib().vvae_maxpool_1x2x2_bwd(_p(x), ldx, _p(dpool), lddp, _p(dskip), ldds, _p(out), out.stride(-2), n * t, h, w, c,
                                       _dt(x), _stream()), "vvae_maxpool_1x2x2_bwd")
    return out


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return maxpool_fwd_raw(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return maxpool_bwd_raw(x, dy.to(x.dtype))


def max_pool_1x2x2(x):
    """nnx.max_pool(x, (1,2,2), strides=(1,2,2)) (reference train/unet.py:50)."""
    return _MaxPool.apply(x)


class _MaxPoolFork(torch.autograd.Function):
    """DownBlock3D's ``return max_pool(x), x`` (reference train/unet.py:50-51) as one node: the second output is x itself, so the
    gradient of the skip connection arrives HERE and is added inside the pool-backward kernel (its ``dskip`` operand, any row
    pitch) instead of by a separate add launch over the largest activations of the network."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return maxpool_fwd_raw(x), x.view_as(x)

    @staticmethod
    def backward(ctx, dpool, dskip):
        (x,) = ctx.saved_tensors
        if dpool is None:
            return dskip
        return maxpool_bwd_raw(x, dpool.to(x.dtype), None if dskip is None else dskip.to(x.dtype))


def max_pool_fork(x):
    """-> (max_pool_1x2x2(x), x) with the skip gradient folded into the pool's backward kernel."""
    return _MaxPoolFork.apply(x)


# --------------------------------------------------------------------------------------------- ConvTranspose (1,2,2)
def _convt_fast(x, cin, cout, ld_in, ld_out):
    return (x.dtype == torch.bfloat16 and not _FORCE_GENERIC[0]
            and lib().vvae_convt_bf16_supported(cin, cout, ld_in, ld_out) == 1)


def _convt_bf16(x, ldx, kernel, bias, out, nt, h, w, cin, cout, dgrad, packed=None):
    """bf16 MFMA ConvTranspose (h, w = low resolution).  dgrad=1: x is dy at 2h x 2w, out is dx at h x w.
    ``packed``: this direction's weights already packed for the whole step (convt_prepack)."""
    if packed is not None:
        ws, wsb, flags = packed, packed.numel(), dgrad | 0x100
    else:
        wsb = lib().vvae_convt_bf16_ws_bytes(cin, cout)
        ws, wsb = _ws(wsb, x.device)
        flags = dgrad
    vox = nt * h * w
    alg = vox * (cin + 4 * cout) * 2
    tag = f"convt_{'dgrad' if dgrad else 'fwd'} {cin}->{cout} @{h}x{w}"
    check(_launch(tag, alg, 2 * vox * 4 * cin * cout, "convt_bf16_kernel",
                  lambda: lib().vvae_convt_1x2x2_bf16(_p(x), ldx, _p(kernel), _p(bias), _p(out), out.stride(-2), nt, h, w, cin, cout,
                                                      flags, _p(ws), wsb, _stream())), "vvae_convt_1x2x2_bf16")
    return out


def convt_prepack(kernels):
    """Pack the forward and input-gradient forms of every ConvTranspose kernel (1, 2, 2, Cin, Cout) of a network in ONE launch
    (vvae_convt_pack_grouped_bf16) -> list of ConvPack (None where the bf16 matrix-core path does not take the layer)."""
    todo = [k for k in kernels if k.is_cuda and k.dtype == torch.float32 and not _FORCE_GENERIC[0]
            and lib().vvae_convt_bf16_supported(k.shape[-2], k.shape[-1], 8, 8) == 1]
    if not todo:
        return [None] * len(kernels)
    sizes = [(lib().vvae_convt_bf16_ws_bytes(k.shape[-2], k.shape[-1]) + 255) // 256 * 256 for k in todo]
    buf = torch.empty(2 * sum(sizes), dtype=torch.uint8, device=todo[0].device)
    packs, ent, off = {}, [], 0
    for k, nb in zip(todo, sizes):
        views = []
        for which in (0, 1):
            v = buf[off:off + nb]
            off += nb
            views.append(v)
            ent.append((k.contiguous(), v, which))
        packs[id(k)] = ConvPack(views[0], views[1])
    n = len(ent)
    VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
    check(lib().vvae_convt_pack_grouped_bf16(VP(*[k.data_ptr() for k, _, _ in ent]), VP(*[v.data_ptr() for _, v, _ in ent]),
                                             IA(*[k.shape[-2] for k, _, _ in ent]), IA(*[k.shape[-1] for k, _, _ in ent]),
                                             IA(*[w for _, _, w in ent]), n, _stream()), "vvae_convt_pack_grouped_bf16")
    return [packs.get(id(k)) for k in kernels]


def convt_fwd_raw(x, kernel, bias, out=None, packed=None):
    x, ldx = rows(x)
    n, t, h, w, cin = x.shape
    cout = kernel.shape[-1]
    if out is None:
        out = torch.empty((n, t, 2 * h, 2 * w, cout), dtype=x.dtype, device=x.device)
    if _convt_fast(x, cin, cout, ldx, out.stride(-2)):
        return _convt_bf16(x, ldx, kernel, bias, out, n * t, h, w, cin, cout, 0, packed)
    check(lib().vvae_convt_1x2x2_fwd(_p(x), ldx, _p(kernel), _p(bias), _p(out), out.stride(-2), n * t, h, w, cin, cout,
                                     _dt(x), _stream()), "vvae_convt_1x2x2_fwd")
    return out


def convt_dgrad_raw(dy, kernel, out=None, packed=None):
    dy, lddy = rows(dy)
    n, t, h2, w2, cout = dy.shape
    cin = kernel.shape[-2]
    if out is None:
        out = torch.empty((n, t, h2 // 2, w2 // 2, cin), dtype=dy.dtype, device=dy.device)
    if _convt_fast(dy, cin, cout, lddy, out.stride(-2)):
        return _convt_bf16(dy, lddy, kernel, None, out, n * t, h2 // 2, w2 // 2, cin, cout, 1, packed)
    check(lib().vvae_convt_1x2x2_dgrad(_p(dy), lddy, _p(kernel), _p(out), out.stride(-2), n * t, h2 // 2, w2 // 2, cin, cout,
                                       _dt(dy), _stream()), "vvae_convt_1x2x2_dgrad")
    return out


def convt_wgrad_raw(x, dy, kshape):
    x, ldx = rows(x)
    dy, lddy = rows(dy)
    n, t, h, w, cin = x.shape
    cout = kshape[-1]
    dw = torch.empty(kshape, dtype=torch.float32, device=x.device)
    ws, wsb = _ws(lib().vvae_convt_1x2x2_wgrad_ws_bytes(n * t, h, w, cin, cout), x.device)
    check(lib().vvae_convt_1x2x2_wgrad(_p(x), ldx, _p(dy), lddy, _p(dw), n * t, h, w, cin, cout, _dt(x), _p(ws), wsb, _stream()),
          "vvae_convt_1x2x2_wgrad")
    return dw


def convt_wgrad_db_raw(x, dy, kshape):
    """-> (dw (1,2,2,Cin,Cout) fp32, db (Cout) fp32): one bf16 matrix-core launch pair where the layer qualifies, else the generic
    weight gradient plus a column sum."""
    x, ldx = rows(x)
    dy, lddy = rows(dy)
    n, t, h, w, cin = x.shape
    cout = kshape[-1]
    if (x.dtype == torch.bfloat16 and not _FORCE_GENERIC[0]
            and lib().vvae_convt_wgrad_bf16_supported(cin, cout, ldx, lddy) == 1):
        dw = torch.empty(kshape, dtype=torch.float32, device=x.device)
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
        wsb = lib().vvae_convt_wgrad_bf16_ws_bytes(n * t, h, w, cin, cout)
        ws, wsb = _ws(wsb, x.device)
        vox = n * t * h * w
        check(_launch(f"convt_wgrad {cin}->{cout} @{h}x{w}", vox * (cin + 4 * cout) * 2, 8 * vox * cin * cout, "convt_wgrad_bf16_kernel",
                      lambda: lib().vvae_convt_1x2x2_wgrad_bf16(_p(x), ldx, _p(dy), lddy, _p(dw), _p(db), n * t, h, w, cin, cout,
                                                                _p(ws), wsb, _stream())), "vvae_convt_1x2x2_wgrad_bf16")
        return dw, db
    return convt_wgrad_raw(x, dy, kshape), colsum_raw(dy)


def colsum_raw(x):
    x, ld = rows(x)
    c = x.shape[-1]
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    v = x.numel() // c
    part = torch.empty((lib().vvae_colsum_blocks(v), c), dtype=torch.float32, device=x.device)
    check(lib().vvae_colsum(_p(x), ld, v, c, _p(out), _p(part), _dt(x), _stream()), "vvae_colsum")
    return out


class _ConvT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, out=None, pack=None):
        k32, b32 = _f32(kernel), _f32(bias)
        ctx.save_for_backward(x, k32)
        ctx.kdtype = kernel.dtype
        ctx.pack = pack
        return convt_fwd_raw(x, k32, b32, out, pack.fwd if pack is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, k32 = ctx.saved_tensors
        dy = dy.to(x.dtype)
        dx = convt_dgrad_raw(dy, k32, packed=ctx.pack.dgrad if ctx.pack is not None else None) if ctx.needs_input_grad[0] else None
        dw, db = convt_wgrad_db_raw(x, dy, tuple(k32.shape))
        return dx, dw.to(ctx.kdtype), db.to(ctx.kdtype), None, None


def conv_transpose_1x2x2(x, kernel, bias, out=None, pack=None):
    """nnx.ConvTranspose((1,2,2), strides (1,2,2)) (reference train/unet.py:61-69).  ``out``: a channel slice of a wider
    NDHWC buffer to write into (see join_channels).  ``pack``: this step's packed weights (convt_prepack)."""
    return _ConvT.apply(x, kernel, bias, out, pack)


class _JoinChannels(torch.autograd.Function):
    """concat([a, b], -1) without the copy (reference train/unet.py:79): a and b ARE the two channel slices of ``buf`` -- their
    producers wrote them there through the kernels' row pitch -- so forward only re-labels the buffer and backward hands each
    producer its slice of the incoming gradient (a pitched view; every consumer takes a row pitch)."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ca, cb = a.shape[-1], b.shape[-1]
        if (buf.shape[-1] != ca + cb or not buf.is_contiguous() or a.data_ptr() != buf.data_ptr()
                or b.data_ptr() != buf.data_ptr() + ca * buf.element_size() or a.stride() != buf.stride() or b.stride() != buf.stride()
                or a.shape[:-1] != buf.shape[:-1] or b.shape[:-1] != buf.shape[:-1]):
            raise VvaeError("join_channels: operands are not the channel slices of the joint buffer")
        ctx.ca = ca
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        return g[..., :ctx.ca], g[..., ctx.ca:], None


def join_channels(a, b, buf):
    return _JoinChannels.apply(a, b, buf)


_ROPE_TABS = {}


def _rope_tabs(cos, sin, dtype):
    """RoPE tables rounded ONCE to the activation dtype (the fused kernels take them pre-rounded; reference layers.py:113-114)."""
    if dtype == torch.float32:
        return cos, sin
    key = (cos.data_ptr(), sin.data_ptr(), dtype)
    hit = _ROPE_TABS.get(key)
    if hit is None or hit[0] is not cos or hit[1] is not sin:
        hit = (cos, sin, cos.to(dtype).to(torch.float32).contiguous(), sin.to(dtype).to(torch.float32).contiguous())
        _ROPE_TABS[key] = hit
    return hit[2], hit[3]


# --------------------------------------------------------------------------------------------- temporal attention core
ATTN_FORCE_GENERIC = [False]      # test hook: use the generic (any head_dim) kernels


class _TemporalAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, q_scale, k_scale, cos, sin, mask, mask_div, heads, eps, inner=1):
        qkv, ld = rows(qkv)
        dt = _dt(qkv)
        c3 = qkv.shape[-1]
        d = c3 // (3 * heads)
        if inner > 1:                      # (b, t, hw, C): sequences stride over frames, no transpose copies
            bsz, t, hw = qkv.shape[:3]
            assert hw == inner
            a = bsz * hw
        else:
            a, t = qkv.shape[:2]
        qs, ks = _f32(q_scale), _f32(k_scale)
        cos, sin = _rope_tabs(cos, sin, qkv.dtype)
        out = torch.empty(qkv.shape[:-1] + (heads * d,), dtype=qkv.dtype, device=qkv.device)
        fast = (not ATTN_FORCE_GENERIC[0]) and lib().vvae_temporal_attn_fast_supported(t, d, ld, heads * d, dt) == 1
        if inner > 1 and not fast:
            raise VvaeError("strided temporal attention needs the lane-per-frame kernels (head_dim 8/16/32/64)")
        lse = None
        nbytes = a * t * heads * d * 4 * qkv.element_size()
        if fast:
            lse = torch.empty((a * heads, t), dtype=torch.float32, device=qkv.device)
            check(_launch(f"temporal_attn_fwd T{t} D{d}", nbytes, 4 * a * heads * t * t * d, "tattn_fwd_fast",
                          lambda: lib().vvae_temporal_attn_fwd_fast(_p(qkv), ld, _p(out), heads * d, _p(lse), _p(qs), _p(ks), _p(cos),
                                                                    _p(sin), _p(mask), mask_div, inner, a, t, heads, d, eps, dt, _stream())),
                  "vvae_temporal_attn_fwd_fast")
        else:
            check(lib().vvae_temporal_attn_fwd(_p(qkv), ld, _p(out), heads * d, _p(qs), _p(ks), _p(cos), _p(sin), _p(mask), mask_div,
                                               a, t, heads, d, eps, dt, _stream()), "vvae_temporal_attn_fwd")
        ctx.qparam, ctx.kparam = q_scale, k_scale
        ctx.save_for_backward(qkv, qs, ks, cos, sin, mask, out, lse)
        ctx.args = (mask_div, heads, eps, q_scale.dtype, fast, inner, a, t)
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, qs, ks, cos, sin, mask, out, lse = ctx.saved_tensors
        mask_div, heads, eps, pdtype, fast, inner, a, t = ctx.args
        qkv, ld = rows(qkv)
        do, lddo = rows(do.to(qkv.dtype))
        c3 = qkv.shape[-1]
        d = c3 // (3 * heads)
        dt = _dt(qkv)
        dqkv = torch.empty(qkv.shape, dtype=qkv.dtype, device=qkv.device)
        if fast:
            nblk = lib().vvae_temporal_attn_fast_blocks(a, t, heads, d, dt)
            part = torch.empty((nblk, 2 * d), dtype=torch.float32, device=qkv.device)
            nbytes = a * t * heads * d * 8 * qkv.element_size()
            check(_launch(f"temporal_attn_bwd T{t} D{d}", nbytes, 10 * a * heads * t * t * d, "tattn_bwd_fast",
                          lambda: lib().vvae_temporal_attn_bwd_fast(_p(qkv), ld, _p(out), heads * d, _p(do), lddo, _p(lse), _p(dqkv), c3,
                                                                    _p(qs), _p(ks), _p(cos), _p(sin), _p(mask), mask_div, inner,
                                                                    _p(part), a, t, heads, d, eps, dt, _stream())),
                  "vvae_temporal_attn_bwd_fast")
            dqs, dks = fold_partials(part, ctx.qparam, ctx.kparam, d)
            if dqs is None:
                return dqkv, None, None, None, None, None, None, None, None, None
        else:
            dqs = torch.empty((d,), dtype=torch.float32, device=qkv.device)
            dks = torch.empty((d,), dtype=torch.float32, device=qkv.device)
            ws, wsb = _ws(lib().vvae_temporal_attn_bwd_ws_bytes(a, heads, d), qkv.device)
            check(lib().vvae_temporal_attn_bwd(_p(qkv), ld, _p(do), lddo, _p(dqkv), c3, _p(qs), _p(ks), _p(cos), _p(sin), _p(mask),
                                               mask_div, _p(dqs), _p(dks), a, t, heads, d, eps, dt, _p(ws), wsb, _stream()),
                  "vvae_temporal_attn_bwd")
        return dqkv, dqs.to(pdtype), dks.to(pdtype), None, None, None, None, None, None, None


def temporal_attention_core(qkv, q_scale, k_scale, cos, sin, mask_u8, mask_div, heads, eps=1e-6, inner=1):
    """q_norm/k_norm -> RoPE -> masked softmax(QK^T/sqrt(D)) V (reference train/layers.py:159-170).

    inner = 1: qkv is (A, T, 3*heads*D), one sequence per leading index.  inner = hw: qkv is (b, T, hw, 3*heads*D) and
    sequence a = b*hw + i strides over frames (the FactoredAttention layout, no transposes).  Output has qkv's leading shape.
    mask_u8: uint8 (ceil(A/mask_div), T), 1 = attend, or None.  cos/sin: fp32 (>=T, D) RoPE tables.
    """
    return _TemporalAttn.apply(qkv, q_scale, k_scale, cos, sin, mask_u8, mask_div, heads, eps, inner)


def temporal_attention_fast_supported(t, d, c3, dtype):
    return dtype in DT and lib().vvae_temporal_attn_fast_supported(t, d, c3, c3 // 3, DT[dtype]) == 1


# --------------------------------------------------------------------------------------------- spatial attention (prep + library core)
def _tok_head(t, a, s):
    """(a, heads, s, d) gradient -> (tensor, token stride, head stride) with one uniform token stride; copies only if it must."""
    if not (t.stride(3) == 1 and t.stride(0) == s * t.stride(2)):
        t = t.transpose(1, 2).contiguous().transpose(1, 2)
    return t, t.stride(2), t.stride(1)


def qk_prep_fwd_raw(qkv, qs, ks, cos, sin, heads, eps=1e-6):
    """qkv (a, s, 3*heads*D) contiguous -> (a, s, 2*heads*D) = [rope(q_norm(q)) | rope(k_norm(k))] (no autograd)."""
    a, s, c3 = qkv.shape
    hd = c3 // 3
    d = hd // heads
    cos, sin = _rope_tabs(cos, sin, qkv.dtype)
    qk = torch.empty((a, s, 2 * hd), dtype=qkv.dtype, device=qkv.device)
    tokens = a * s
    check(_launch(f"qk_prep_fwd S{s} D{d}", tokens * 4 * hd * qkv.element_size(), 0, "qk_prep_fwd_kernel",
                  lambda: lib().vvae_qk_prep_fwd(_p(qkv), c3, _p(qk), 2 * hd, _p(qs), _p(ks), _p(cos), _p(sin), tokens, s, heads, d,
                                                 eps, _dt(qkv), _stream())), "vvae_qk_prep_fwd")
    return qk


def qk_prep_bwd_raw(qkv, dq, dk, dv, qs, ks, cos, sin, heads, eps=1e-6, params=None):
    """dq, dk, dv: (a, heads, s, D) gradients (any strides) -> (dqkv (a, s, 3*heads*D), dq_scale (D), dk_scale (D)) (no autograd)."""
    a, s, c3 = qkv.shape
    hd = c3 // 3
    d = hd // heads
    dq, dq_ts, dq_hs = _tok_head(dq, a, s)
    dk, dk_ts, dk_hs = _tok_head(dk, a, s)
    dv, dv_ts, dv_hs = _tok_head(dv, a, s)
    cos, sin = _rope_tabs(cos, sin, qkv.dtype)
    tokens = a * s
    dqkv = torch.empty((a, s, c3), dtype=qkv.dtype, device=qkv.device)
    nblk = lib().vvae_qk_prep_blocks(tokens, heads, d)
    part = torch.empty((nblk, 2, d), dtype=torch.float32, device=qkv.device)
    check(_launch(f"qk_prep_bwd S{s} D{d}", tokens * 8 * hd * qkv.element_size(), 0, "qk_prep_bwd_kernel",
                  lambda: lib().vvae_qk_prep_bwd(_p(qkv), c3, _p(dq), dq_ts, dq_hs, _p(dk), dk_ts, dk_hs, _p(dv), dv_ts, dv_hs,
                                                 _p(dqkv), c3, _p(qs), _p(ks), _p(cos), _p(sin), _p(part), tokens, s, heads, d, eps,
                                                 _dt(qkv), _stream())), "vvae_qk_prep_bwd")
    g0, g1 = fold_partials(part, params[0], params[1], d) if params is not None else fold_partials(part, None, None, d)
    return dqkv, g0, g1


SPATIAL_FORCE_LIBRARY_CORE = [False]      # test hook: keep the prep kernels + library flash core even where the fused kernels apply


def spatial_attn_fused_supported(qkv, heads):
    a, s, c3 = qkv.shape
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16
            and lib().vvae_spatial_attn_supported(s, c3 // (3 * heads), DT[qkv.dtype]) == 1)


def spatial_attn_fwd_raw(qkv, qs, ks, cos, sin, heads, eps=1e-6):
    """Fused q/k-norm + RoPE + attention forward on (a, s, 3*heads*64) bf16 -> (out (a, s, heads*64), lse2 (a*heads, s)) (no autograd)."""
    a, s, c3 = qkv.shape
    hd = c3 // 3
    d = hd // heads
    cos, sin = _rope_tabs(cos, sin, qkv.dtype)
    out = torch.empty((a, s, hd), dtype=qkv.dtype, device=qkv.device)
    lse2 = torch.empty((a * heads, s), dtype=torch.float32, device=qkv.device)
    nbytes = a * s * (c3 + hd) * qkv.element_size()
    check(_launch(f"spatial_attn_fwd S{s} D{d}", nbytes, 4 * a * heads * s * s * d, "sattn_fwd_kernel",
                  lambda: lib().vvae_spatial_attn_fwd(_p(qkv), c3, _p(out), hd, _p(lse2), _p(qs), _p(ks), _p(cos), _p(sin), a, s, heads, d,
                                                      eps, _dt(qkv), _stream())), "vvae_spatial_attn_fwd")
    return out, lse2


def spatial_attn_bwd_raw(qkv, out, lse2, dout, qs, ks, cos, sin, heads, eps=1e-6, params=None):
    """-> (dqkv (a, s, 3*heads*64), dq_scale (64), dk_scale (64)) of the fused spatial attention (no autograd)."""
    a, s, c3 = qkv.shape
    hd = c3 // 3
    d = hd // heads
    cos, sin = _rope_tabs(cos, sin, qkv.dtype)
    dqkv = torch.empty((a, s, c3), dtype=qkv.dtype, device=qkv.device)
    part = torch.empty((a * heads, 2, d), dtype=torch.float32, device=qkv.device)
    nbytes = a * s * (2 * c3 + 2 * hd) * qkv.element_size()
    check(_launch(f"spatial_attn_bwd S{s} D{d}", nbytes, 14 * a * heads * s * s * d, "sattn_bwd_kernel",
                  lambda: lib().vvae_spatial_attn_bwd(_p(qkv), c3, _p(out), hd, _p(dout), hd, _p(lse2), _p(dqkv), c3, _p(qs), _p(ks),
                                                      _p(cos), _p(sin), _p(part), a, s, heads, d, eps, _dt(qkv), _stream())),
          "vvae_spatial_attn_bwd")
    g0, g1 = fold_partials(part, params[0], params[1], d) if params is not None else fold_partials(part, None, None, d)
    return dqkv, g0, g1


class _SpatialAttnFused(torch.autograd.Function):
    """q/k-norm + RoPE + softmax(QK^T/sqrt(D))V, one hand-written kernel per direction (reference train/layers.py:153-170, :217-221)."""

    @staticmethod
    def forward(ctx, qkv, q_scale, k_scale, cos, sin, heads, eps):
        qkv = qkv.contiguous()
        qs, ks = _f32(q_scale), _f32(k_scale)
        out, lse2 = spatial_attn_fwd_raw(qkv, qs, ks, cos, sin, heads, eps)
        ctx.save_for_backward(qkv, qs, ks, cos, sin, out, lse2)
        ctx.misc = (heads, eps, q_scale.dtype)
        ctx.qparam, ctx.kparam = q_scale, k_scale
        return out

    @staticmethod
    def backward(ctx, do):
        qkv, qs, ks, cos, sin, out, lse2 = ctx.saved_tensors
        heads, eps, pdtype = ctx.misc
        do = do.to(qkv.dtype).contiguous()
        dqkv, dqs, dks = spatial_attn_bwd_raw(qkv, out, lse2, do, qs, ks, cos, sin, heads, eps, (ctx.qparam, ctx.kparam))
        if dqs is None:
            return dqkv, None, None, None, None, None, None
        return dqkv, dqs.to(pdtype), dks.to(pdtype), None, None, None, None


class _SpatialAttn(torch.autograd.Function):
    """q/k-norm + RoPE prep (HIP, one launch) -> library flash-attention core -> (backward) core -> prep backward (HIP, one launch).

    The custom Function owns the whole span so autograd never materialises the slice / cat / transpose gradients of the
    q, k, v views (reference train/layers.py:153-170 with the mask-free spatial call of :217-221).
    """

    @staticmethod
    def _views(qkv, qk, heads):
        a, s, c3 = qkv.shape
        hd = c3 // 3
        d = hd // heads
        q = qk[..., :hd].view(a, s, heads, d).transpose(1, 2)
        k = qk[..., hd:].view(a, s, heads, d).transpose(1, 2)
        v = qkv[..., 2 * hd:].view(a, s, heads, d).transpose(1, 2)
        return q, k, v

    @staticmethod
    def forward(ctx, qkv, q_scale, k_scale, cos, sin, heads, eps):
        qkv = qkv.contiguous()
        a, s, c3 = qkv.shape
        qs, ks = _f32(q_scale), _f32(k_scale)
        qk = qk_prep_fwd_raw(qkv, qs, ks, cos, sin, heads, eps)
        q, k, v = _SpatialAttn._views(qkv, qk, heads)
        out, lse, cq, ck, mq, mk, seed, off, _ = torch.ops.aten._scaled_dot_product_flash_attention(q, k, v, 0.0, False, False)
        ctx.save_for_backward(qkv, qs, ks, cos, sin, qk, out, lse, seed, off)
        ctx.misc = (cq, ck, mq, mk, heads, eps, q_scale.dtype)
        ctx.qparam, ctx.kparam = q_scale, k_scale
        return out.transpose(1, 2).reshape(a, s, c3 // 3)

    @staticmethod
    def backward(ctx, do):
        qkv, qs, ks, cos, sin, qk, out, lse, seed, off = ctx.saved_tensors
        cq, ck, mq, mk, heads, eps, pdtype = ctx.misc
        a, s, c3 = qkv.shape
        q, k, v = _SpatialAttn._views(qkv, qk, heads)
        do4 = do.to(qkv.dtype).reshape(a, s, heads, c3 // (3 * heads)).transpose(1, 2)
        dq, dk, dv = torch.ops.aten._scaled_dot_product_flash_attention_backward(do4, q, k, v, out, lse, cq, ck, mq, mk, 0.0, False,
                                                                                 seed, off)
        dqkv, dqs, dks = qk_prep_bwd_raw(qkv, dq, dk, dv, qs, ks, cos, sin, heads, eps, (ctx.qparam, ctx.kparam))
        if dqs is None:
            return dqkv, None, None, None, None, None, None
        return dqkv, dqs.to(pdtype), dks.to(pdtype), None, None, None, None


def spatial_attention_supported(qkv, heads, max_len):
    """bf16 GPU tensors with a head_dim the prep kernels take (fp32 and masked calls stay on the composed path)."""
    if not (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.shape[1] <= max_len):
        return False
    d = qkv.shape[-1] // (3 * heads)
    return lib().vvae_qk_prep_supported(d, DT[qkv.dtype]) == 1


def spatial_attention_core(qkv, q_scale, k_scale, cos, sin, heads, eps=1e-6):
    """q_norm/k_norm -> RoPE -> softmax(QK^T/sqrt(D)) V over (a, s, 3*heads*D), no mask (reference train/layers.py:153-170)."""
    if spatial_attn_fused_supported(qkv, heads) and not SPATIAL_FORCE_LIBRARY_CORE[0]:
        return _SpatialAttnFused.apply(qkv, q_scale, k_scale, cos, sin, heads, eps)
    if not SPATIAL_FORCE_LIBRARY_CORE[0]:
        note_fallback(("sattn-core", tuple(qkv.shape[1:]), heads),
                      f"spatial attention over (s, 3 h d) = {tuple(qkv.shape[1:])}, {heads} heads: the fused HIP kernel takes head_dim 64, s <= 256; "
                      "the library flash-attention core runs between two HIP prep launches instead")
    return _SpatialAttn.apply(qkv, q_scale, k_scale, cos, sin, heads, eps)


# --------------------------------------------------------------------------------------------- reparameterise + KL
def _loss_part(b, m, device):
    """Scratch for the per-workgroup partial sums of a per-sample loss reduction (folded in fixed order: no float atomics)."""
    return torch.empty((lib().vvae_loss_part_floats(b, m),), dtype=torch.float32, device=device)


class _ReparamKl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, logvar, eps, mask, want_z, want_kl):
        mean, logvar = mean.contiguous(), logvar.contiguous()
        b, t = mean.shape[0], mean.shape[1]
        per = mean.numel() // (b * t)
        z = torch.empty(mean.shape, dtype=torch.float32, device=mean.device) if want_z else None
        kl = torch.empty((b,), dtype=torch.float32, device=mean.device) if want_kl else None
        eps = eps.to(torch.float32).contiguous() if eps is not None else None
        mask = mask.to(torch.float32).contiguous() if mask is not None else None
        part = _loss_part(b, t * per, mean.device) if want_kl else None
        check(lib().vvae_reparam_kl_fwd(_p(mean), _p(logvar), _p(eps), _p(mask), _p(z), _p(kl), _p(part), b, t, per, _dt(mean),
                                        _stream()), "vvae_reparam_kl_fwd")
        ctx.save_for_backward(mean, logvar, eps, mask)
        ctx.flags = (want_z, want_kl)
        outs = tuple(o for o in (z, kl) if o is not None)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        mean, logvar, eps, mask = ctx.saved_tensors
        want_z, want_kl = ctx.flags
        grads = list(grads)
        dz = grads.pop(0) if want_z else None
        gkl = grads.pop(0) if want_kl else None
        dz = dz.to(torch.float32).contiguous() if dz is not None else None
        gkl = gkl.to(torch.float32).contiguous() if gkl is not None else None
        b, t = mean.shape[0], mean.shape[1]
        per = mean.numel() // (b * t)
        dmean, dlogvar = torch.empty_like(mean), torch.empty_like(logvar)
        check(lib().vvae_reparam_kl_bwd(_p(mean), _p(logvar), _p(eps), _p(mask), _p(dz), _p(gkl), _p(dmean), _p(dlogvar), b, t,
                                        per, _dt(mean), _stream()), "vvae_reparam_kl_bwd")
        return dmean, dlogvar, None, None, None, None


def reparameterise(mean, logvar, eps):
    """z = mean + eps * exp(logvar / 2), fp32 out (reference train/model.py:124-128)."""
    return _ReparamKl.apply(mean, logvar, eps, None, True, False)


def kl_per_sample(mean, logvar, mask_bt):
    """mean_{t,hw,c}[0.5 (e^lv - 1 - lv + mu^2) m_t / len] per sample (reference train/rl_nonadversarial.py:146-147)."""
    return _ReparamKl.apply(mean, logvar, None, mask_bt, False, True)


def reparameterise_kl(mean, logvar, eps, mask_bt):
    """Both in one pass over (mean, logvar): -> (z, kl_per_sample)."""
    return _ReparamKl.apply(mean, logvar, eps, mask_bt, True, True)


# --------------------------------------------------------------------------------------------- masked MSE / MAE
class _MaskedMseMae(torch.autograd.Function):
    @staticmethod
    def forward(ctx, video, recon, mask, video_div, partials=False):
        recon = recon.contiguous()
        video = video.to(recon.dtype).contiguous()
        mask = mask.to(torch.float32).contiguous()
        b, t = recon.shape[0], recon.shape[1]
        p = recon.numel() // (b * t)
        assert video.shape[0] * video_div == b and video.shape[1:] == recon.shape[1:]
        part = _loss_part(b, t * p, recon.device)
        if partials:                                     # (b, chunks) per-workgroup partial sums: the loss tail adds them up, no fold launch
            mse = mae = None
        else:
            mse = torch.empty((b,), dtype=torch.float32, device=recon.device)
            mae = torch.empty((b,), dtype=torch.float32, device=recon.device)
        check(lib().vvae_masked_mse_mae_fwd(_p(video), _p(recon), _p(mask), _p(mse), _p(mae), _p(part), b, t, p, video_div,
                                            _dt(recon), _stream()), "vvae_masked_mse_mae_fwd")
        ctx.save_for_backward(video, recon, mask)
        ctx.video_div = video_div
        ctx.set_materialize_grads(False)                 # an unused output's gradient stays None (the kernel takes NULL), no zero fill
        if partials:
            chunks = part.numel() // (2 * b)
            return part[:b * chunks].view(b, chunks), part[b * chunks:].view(b, chunks)
        return mse, mae

    @staticmethod
    def backward(ctx, gmse, gmae):
        video, recon, mask = ctx.saved_tensors
        if gmse is None and gmae is None:
            return None, None, None, None, None
        b, t = recon.shape[0], recon.shape[1]
        p = recon.numel() // (b * t)
        # partial-sum outputs: every partial of a sample carries the sample's gradient (a stride-0 expansion, what a plain sum over the
        # partials or the loss tails hand back): column 0 is it.  Anything else (partials weighted differently) is not what this op supports.
        for gg in (gmse, gmae):
            if gg is not None and gg.dim() == 2 and gg.shape[1] > 1 and gg.stride(1) != 0:
                raise VvaeError("masked_mse_mae(partials=True): the partial sums of a sample must enter the loss through their plain sum")
        gmse = (gmse[:, 0] if gmse.dim() == 2 else gmse).to(torch.float32).contiguous() if gmse is not None else None
        gmae = (gmae[:, 0] if gmae.dim() == 2 else gmae).to(torch.float32).contiguous() if gmae is not None else None
        dr = torch.empty_like(recon)
        check(lib().vvae_masked_mse_mae_bwd(_p(video), _p(recon), _p(mask), _p(gmse), _p(gmae), _p(dr), b, t, p, ctx.video_div,
                                            _dt(recon), _stream()), "vvae_masked_mse_mae_bwd")
        return None, dr, None, None, None


def masked_mse_mae(video, recon, mask_bt, video_div=1, partials=False):
    """Per-sample masked MSE and MAE (reference train/rl_nonadversarial.py:114-121); gradient flows to recon only.
    ``partials``: return each as (b, chunks) partial sums instead (their row sums are the per-sample values; ops.plain_loss_tail takes them)."""
    return _MaskedMseMae.apply(video, recon, mask_bt, video_div, partials)


_UNIT_GRAD = {}


def unit_grad(like):
    """A cached fp32 scalar 1.0 on ``like``'s device: the root gradient of ``loss.backward`` / ``torch.autograd.grad(loss, ...)``.  Handing it
    over explicitly (``gradient=`` / ``grad_outputs=``) spares the engine's ``ones_like`` fill launch, and a backward that recognises it by
    its storage (_PlainLossTail) spares the multiplication by it: two ~5 us launches of every replayed step."""
    key = (like.device.type, like.device.index)
    t = _UNIT_GRAD.get(key)
    if t is None:
        t = _UNIT_GRAD[key] = torch.ones((), dtype=torch.float32, device=like.device)
    return t


class _PlainLossTail(torch.autograd.Function):
    """The scalar end of loss.loss_fn_plain in one launch, gradients included (vvae_loss_tail_plain)."""

    @staticmethod
    def forward(ctx, mse_ps, kl_ps, selection, mask, max_rate, magnify, gamma1, gamma2):
        b, t = mask.shape
        out = torch.empty((5,), dtype=torch.float32, device=mse_ps.device)
        grads = torch.empty((2 * b + b * t,), dtype=torch.float32, device=mse_ps.device)
        sel = selection.reshape(b, t).to(torch.float32).contiguous()
        kl_cols = kl_ps.numel() // b                     # (b,) or (b, k) partial sums of the per-sample term (ops.encoder_head: one per frame)
        mse_cols = mse_ps.numel() // b                   # likewise (ops.masked_mse_mae(partials=True): one per workgroup)
        check(lib().vvae_loss_tail_plain(_p(mse_ps.contiguous()), mse_cols, _p(kl_ps.contiguous()), kl_cols, _p(sel), _p(mask), b, t, float(max_rate),
                                         float(magnify), float(gamma1), float(gamma2), _p(out), _p(grads), _stream()),
              "vvae_loss_tail_plain")
        ctx.save_for_backward(grads)
        ctx.bt, ctx.sel_shape, ctx.sel_dtype, ctx.kl_shape, ctx.mse_shape = (b, t), selection.shape, selection.dtype, kl_ps.shape, mse_ps.shape
        aux = out[1:]
        ctx.mark_non_differentiable(aux)
        ctx.set_materialize_grads(False)
        return out[0], aux

    @staticmethod
    def backward(ctx, go, _gaux):
        if go is None:
            return (None,) * 8
        (grads,) = ctx.saved_tensors
        b, t = ctx.bt
        unit = _UNIT_GRAD.get((go.device.type, go.device.index))
        # one launch for the three gradients -- none when the root gradient is the constant 1.0 of unit_grad (same storage)
        g = grads if (unit is not None and go.data_ptr() == unit.data_ptr() and go.numel() == 1) else grads * go
        gmse, gkl = g[:b], g[b:2 * b]
        if len(ctx.kl_shape) == 2:                       # every partial sum of a sample has the sample's gradient: a stride-0 view, no launch
            gkl = gkl.unsqueeze(1).expand(ctx.kl_shape)
        if len(ctx.mse_shape) == 2:
            gmse = gmse.unsqueeze(1).expand(ctx.mse_shape)
        return gmse, gkl, g[2 * b:].view(ctx.sel_shape).to(ctx.sel_dtype), None, None, None, None, None


def plain_loss_tail_ok(mse_ps, kl_ps, selection, mask):
    return (mse_ps.is_cuda and mse_ps.dtype == torch.float32 and kl_ps.dtype == torch.float32 and mask.dtype == torch.float32
            and mask.dim() == 2 and mask.is_contiguous() and mask.shape[0] <= 1024 and selection.numel() == mask.numel()
            and kl_ps.dim() in (1, 2) and kl_ps.shape[0] == mask.shape[0] and mse_ps.dim() in (1, 2) and mse_ps.shape[0] == mask.shape[0])


def plain_loss_tail(mse_ps, kl_ps, selection, mask, hparams):
    """-> (loss, (MSE, selection_loss, kl_loss, kept_frame_density)) of loss.loss_fn_plain from the per-sample sums."""
    loss, aux = _PlainLossTail.apply(mse_ps, kl_ps, selection, mask, hparams["max_compression_rate"], hparams["magnify_negatives_rate"],
                                     hparams["gamma1"], hparams["gamma2"])
    return loss, aux.unbind(0)


class _RlLossTail(torch.autograd.Function):
    """The scalar end of loss.loss_fn (the rl flavour's pair / REINFORCE loss) in one launch, gradients included (vvae_loss_tail_rl)."""

    @staticmethod
    def forward(ctx, mse, mae, perc, kl, selection, actions, mask, max_rate, magnify, g1, g2, g3, g4, w):
        b2, t = mask.shape
        out = torch.empty((9,), dtype=torch.float32, device=mask.device)
        grads = torch.empty((4 * b2 + b2 * t,), dtype=torch.float32, device=mask.device)
        sel = selection.reshape(b2, t).to(torch.float32).contiguous()
        act = actions.reshape(b2, t).to(torch.float32).contiguous()
        cols = mse.numel() // b2
        kl_cols = kl.numel() // b2                       # (2b,) or (2b, k) partial sums of the per-sample term (ops.encoder_head_rl: one per frame)
        pc = perc.to(torch.float32).contiguous() if perc is not None else None
        check(lib().vvae_loss_tail_rl(_p(mse.contiguous()), _p(mae.contiguous()), cols, _p(pc), _p(kl.to(torch.float32).contiguous()), kl_cols, _p(sel), _p(act),
                                      _p(mask), b2, t, float(max_rate), float(magnify), float(g1), float(g2), float(g3), float(g4), float(w), _p(out),
                                      _p(grads), _stream()), "vvae_loss_tail_rl")
        ctx.save_for_backward(grads)
        ctx.meta = (b2, t, mse.shape, mae.shape, perc is not None, selection.shape, selection.dtype, kl.shape)
        aux = out[1:]
        ctx.mark_non_differentiable(aux)
        ctx.set_materialize_grads(False)
        return out[0], aux

    @staticmethod
    def backward(ctx, go, _gaux):
        if go is None:
            return (None,) * 14
        (grads,) = ctx.saved_tensors
        b2, t, mse_shape, mae_shape, has_perc, sel_shape, sel_dtype, kl_shape = ctx.meta
        unit = _UNIT_GRAD.get((go.device.type, go.device.index))
        g = grads if (unit is not None and go.data_ptr() == unit.data_ptr() and go.numel() == 1) else grads * go
        gmse, gmae = g[:b2], g[b2:2 * b2]
        if len(mse_shape) == 2:
            gmse = gmse.unsqueeze(1).expand(mse_shape)
        if len(mae_shape) == 2:
            gmae = gmae.unsqueeze(1).expand(mae_shape)
        gkl = g[3 * b2:4 * b2]
        if len(kl_shape) == 2:                           # every partial sum of a sample has the sample's gradient: a stride-0 view, no launch
            gkl = gkl.unsqueeze(1).expand(kl_shape)
        return (gmse, gmae, g[2 * b2:3 * b2] if has_perc else None, gkl, g[4 * b2:].view(sel_shape).to(sel_dtype), None, None,
                None, None, None, None, None, None, None)


def rl_loss_tail_ok(mse, mae, kl, selection, actions, mask):
    return (mse.is_cuda and all(x.dtype == torch.float32 for x in (mse, mae, kl, mask)) and mask.dim() == 2 and mask.is_contiguous()
            and mask.shape[0] % 2 == 0 and mask.shape[0] <= 1024 and selection.numel() == mask.numel() and actions.numel() == mask.numel()
            and mse.dim() in (1, 2) and mse.shape == mae.shape and mse.shape[0] == mask.shape[0] and kl.dim() in (1, 2) and kl.shape[0] == mask.shape[0])


def rl_loss_tail(mse, mae, perc, kl, selection, actions, mask, hparams):
    """-> (loss, (MSE, perceptual_loss, selection_loss, kl_loss, kept_frame_density, mean_trajectory_prob, rl_loss, per_sample_MAE)) of loss.loss_fn from
    the per-sample sums (``mse`` / ``mae`` may be (2b, chunks) partial sums), the selection probabilities, the sampled actions and the mask."""
    loss, aux = _RlLossTail.apply(mse, mae, perc, kl, selection, actions, mask, hparams["max_compression_rate"], hparams["magnify_negatives_rate"],
                                  hparams["gamma1"], hparams["gamma2"], hparams["gamma3"], hparams["gamma4"], hparams["rl_loss_weight"])
    return loss, aux.unbind(0)


# --------------------------------------------------------------------------------------------- encoder heads + latent gate
class _EncoderHead(torch.autograd.Function):
    """log-variance, selection logits, Gumbel-sigmoid STE, reparameterisation, per-frame KL and the latent gate of the model.py flavour in
    ONE launch each way (vvae_encoder_head_fwd / _bwd; reference train/model.py:53-59,121-133, train/layers.py:226-252).  The parameter
    gradients leave the backward kernel as one partial row per frame and join the grouped folds of ``deferred_wgrad`` (fold_partials)."""

    @staticmethod
    def forward(ctx, mean, v, w1, b1, w2, b2, fill, u, eps, mask_bt):
        b, t, hw, ld = mean.shape
        mean, v = mean.contiguous(), v.contiguous()
        dev = mean.device
        w1f, b1f, w2f, b2f, ff = _f32(w1), _f32(b1), _f32(w2), _f32(b2), _f32(fill)
        u = u.to(torch.float32).contiguous()
        eps = eps.to(torch.float32).contiguous()
        if mask_bt.dtype != torch.float32 or (mask_bt.shape[1] > 1 and mask_bt.stride(1) != 1):
            mask_bt = mask_bt.to(torch.float32).contiguous()
        logvar, comp = torch.empty_like(mean), torch.empty_like(mean)
        sel = torch.empty((b, t, 1, 1), dtype=torch.float32, device=dev)
        y = torch.empty((b * t,), dtype=torch.float32, device=dev)
        s1 = torch.empty((b * t, hw), dtype=torch.float32, device=dev)
        kl = torch.empty((b, t), dtype=torch.float32, device=dev)
        mp = mask_bt.stride(0) if mask_bt.shape[0] > 1 else 0
        check(lib().vvae_encoder_head_fwd(_p(mean), _p(v), _p(w1f), _p(b1f), _p(w2f), _p(b2f), _p(u), _p(eps), _p(mask_bt), mp,
                                          _p(ff), _p(logvar), _p(comp), _p(sel), _p(y), _p(s1), _p(kl), b, t, hw, ld, _stream()),
              "vvae_encoder_head_fwd")
        ctx.save_for_backward(mean, v, eps, mask_bt, ff, w1f, w2f, y, s1, sel, logvar)
        ctx.params = (w1, b1, w2, b2, fill)
        ctx.set_materialize_grads(False)
        return logvar, comp, sel, kl

    @staticmethod
    def backward(ctx, dlv, dcomp, dsel, gkl):
        mean, v, eps, mask_bt, ff, w1f, w2f, y, s1, sel, logvar = ctx.saved_tensors
        w1, b1, w2, b2, fill = ctx.params
        b, t, hw, ld = mean.shape
        dev = mean.device
        if dcomp is not None:
            dcomp = dcomp.to(torch.bfloat16).contiguous()
        if dlv is not None:
            dlv = dlv.to(torch.bfloat16).contiguous()
        if dsel is not None:
            dsel = dsel.to(torch.float32).contiguous()
        gb = gt = 0
        if gkl is not None:
            if gkl.dtype != torch.float32:
                gkl = gkl.to(torch.float32)
            gb, gt = gkl.stride(0), gkl.stride(1)
        dmean, dv = torch.empty_like(mean), torch.empty_like(v)
        f = b * t
        part = torch.empty((2 * f * ld + f * hw + 8 * f,), dtype=torch.float32, device=dev)      # row widths: multiples of four floats (the folds)
        p1, p2, p3 = part[:f * ld].view(f, ld), part[f * ld:f * (ld + hw)].view(f, hw), part[f * (ld + hw):f * (2 * ld + hw)].view(f, ld)
        pb = part[f * (2 * ld + hw):].view(2, f, 4)
        mp = mask_bt.stride(0) if mask_bt.shape[0] > 1 else 0
        check(lib().vvae_encoder_head_bwd(_p(mean), _p(v), _p(logvar), _p(eps), _p(mask_bt), mp, _p(ff), _p(w1f), _p(w2f), _p(y), _p(s1), _p(sel),
                                          _p(dcomp), _p(dsel), _p(gkl), gb, gt, _p(dlv), _p(dmean), _p(dv), _p(p1), _p(p2), _p(p3), _p(pb), b, t, hw,
                                          ld, _stream()), "vvae_encoder_head_bwd")
        dw1, _ = fold_partials(p1, w1, None, ld)
        dw2, _ = fold_partials(p2, w2, None, hw)
        dfill, _ = fold_partials(p3, fill, None, ld)
        db1, _ = fold_partials(pb[0], b1, None, 1)
        db2, _ = fold_partials(pb[1], b2, None, 1)
        like = lambda g, p: None if g is None else g.reshape(p.shape).to(p.dtype)
        return (dmean, dv, like(dw1, w1), like(db1, b1), like(dw2, w2), like(db2, b2), like(dfill, fill), None, None, None)


def encoder_head_ok(mean, v, w1, b1, w2, b2, fill):
    """bf16 GPU activations (b, t, hw, ld), fp32 parameters of the shapes nnx.Linear(ld, 1) / nnx.Linear(hw, 1) give, ld a multiple of 8."""
    if not (mean.is_cuda and mean.dim() == 4 and mean.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and v.shape == mean.shape):
        return False
    b, t, hw, ld = mean.shape
    return (w1.numel() == ld and b1.numel() == 1 and w2.numel() == hw and b2.numel() == 1 and fill.numel() == ld
            and all(p.dtype == torch.float32 for p in (w1, b1, w2, b2, fill)) and bool(lib().vvae_encoder_head_ok(b, t, hw, ld)))


def encoder_head(mean, v, w1, b1, w2, b2, fill, u, eps, mask_bt):
    """-> (log_variance bf16, compressed_representation bf16, selection (b, t, 1, 1) in {0, 1} with the straight-through gradient,
    kl (b, t): the per-sample KL term as one partial sum per frame)."""
    return _EncoderHead.apply(mean, v, w1, b1, w2, b2, fill, u, eps, mask_bt)


class _EncoderHeadRl(torch.autograd.Function):
    """The rl flavour's heads + reparameterisation + KL + pair doubling + Bernoulli frame masks + latent gate in ONE launch each way
    (vvae_encoder_head_rl_fwd / _bwd; reference train/rl_model.py:50-60,119-147): what rl_model.VideoVAE.forward did with softplus, log, two
    addmm, add, sigmoid, ops.reparameterise_kl, five repeat_interleave and ops.rl_gate (and their backward launches)."""

    @staticmethod
    def forward(ctx, mean, v, w1, b1, w2, b2, fill, u2, eps, mask_bt):
        b, t, hw, ld = mean.shape
        mean, v = mean.contiguous(), v.contiguous()
        dev = mean.device
        w1f, b1f, w2f, b2f, ff = _f32(w1), _f32(b1), _f32(w2), _f32(b2), _f32(fill)
        u2 = u2.reshape(2 * b, t).to(torch.float32).contiguous()
        eps = eps.to(torch.float32).contiguous()
        if mask_bt.dtype != torch.float32 or (mask_bt.shape[1] > 1 and mask_bt.stride(1) != 1):
            mask_bt = mask_bt.to(torch.float32).contiguous()
        shape2 = (2 * b, t, hw, ld)
        logvar2, mean2, comp2 = (torch.empty(shape2, dtype=mean.dtype, device=dev) for _ in range(3))
        sel2 = torch.empty((2 * b, t, 1, 1), dtype=torch.float32, device=dev)             # the pair-doubled probability
        mask2 = torch.empty((2 * b, t, 1, 1), dtype=torch.float32, device=dev)
        y = torch.empty((b * t,), dtype=torch.float32, device=dev)
        s1 = torch.empty((b * t, hw), dtype=torch.float32, device=dev)
        kl2 = torch.empty((2 * b, t), dtype=torch.float32, device=dev)
        mp = mask_bt.stride(0) if mask_bt.shape[0] > 1 else 0
        check(lib().vvae_encoder_head_rl_fwd(_p(mean), _p(v), _p(w1f), _p(b1f), _p(w2f), _p(b2f), _p(u2), _p(eps), _p(mask_bt), mp, _p(ff),
                                             _p(logvar2), _p(mean2), _p(comp2), _p(sel2), _p(mask2), _p(y), _p(s1), _p(kl2), b, t, hw, ld, _stream()),
              "vvae_encoder_head_rl_fwd")
        ctx.save_for_backward(mean, v, eps, mask_bt, ff, w1f, w2f, y, s1, mask2, logvar2)
        ctx.params = (w1, b1, w2, b2, fill)
        ctx.mark_non_differentiable(mask2)
        ctx.set_materialize_grads(False)
        return logvar2, mean2, comp2, sel2, mask2, kl2

    @staticmethod
    def backward(ctx, dlv2, dmean2, dcomp2, dsel2, _dmask2, gkl):
        mean, v, eps, mask_bt, ff, w1f, w2f, y, s1, mask2, logvar2 = ctx.saved_tensors
        w1, b1, w2, b2, fill = ctx.params
        b, t, hw, ld = mean.shape
        dev = mean.device
        if dcomp2 is not None:
            dcomp2 = dcomp2.to(torch.bfloat16).contiguous()
        if dsel2 is not None:
            dsel2 = dsel2.reshape(2 * b, t).to(torch.float32).contiguous()
        # gradients arriving at the pair-doubled log-variance / mean copies (the loss reads them only through kl2; a caller that does
        # use them pays two small framework launches)
        dlv_ext = None if dlv2 is None else dlv2.reshape(b, 2, t, hw, ld).sum(1).to(torch.bfloat16).contiguous()
        gb = gt = 0
        if gkl is not None:
            if gkl.dtype != torch.float32:
                gkl = gkl.to(torch.float32)
            gb, gt = gkl.stride(0), gkl.stride(1)
        dmean, dv = torch.empty_like(mean), torch.empty_like(v)
        f = b * t
        part = torch.empty((2 * f * ld + f * hw + 8 * f,), dtype=torch.float32, device=dev)
        p1, p2, p3 = part[:f * ld].view(f, ld), part[f * ld:f * (ld + hw)].view(f, hw), part[f * (ld + hw):f * (2 * ld + hw)].view(f, ld)
        pb = part[f * (2 * ld + hw):].view(2, f, 4)
        mp = mask_bt.stride(0) if mask_bt.shape[0] > 1 else 0
        check(lib().vvae_encoder_head_rl_bwd(_p(mean), _p(v), _p(logvar2), _p(eps), _p(mask_bt), mp, _p(ff), _p(w1f), _p(w2f), _p(y), _p(s1), _p(mask2),
                                             _p(dcomp2), _p(dsel2), _p(gkl), gb, gt, _p(dlv_ext), _p(dmean), _p(dv), _p(p1), _p(p2), _p(p3), _p(pb), b, t,
                                             hw, ld, _stream()), "vvae_encoder_head_rl_bwd")
        if dmean2 is not None:
            dmean = dmean + dmean2.reshape(b, 2, t, hw, ld).sum(1).to(dmean.dtype)
        dw1, _ = fold_partials(p1, w1, None, ld)
        dw2, _ = fold_partials(p2, w2, None, hw)
        dfill, _ = fold_partials(p3, fill, None, ld)
        db1, _ = fold_partials(pb[0], b1, None, 1)
        db2, _ = fold_partials(pb[1], b2, None, 1)
        like = lambda g, p: None if g is None else g.reshape(p.shape).to(p.dtype)
        return (dmean, dv, like(dw1, w1), like(db1, b1), like(dw2, w2), like(db2, b2), like(dfill, fill), None, None, None)


def encoder_head_rl(mean, v, w1, b1, w2, b2, fill, u2, eps, mask_bt):
    """-> (log_variance, mean, compressed_representation: bf16 (2b, t, hw, ld), pair-doubled; selection probability (2b, t, 1, 1) fp32;
    selection_mask (2b, t, 1, 1) in {0, 1}; kl (2b, t): the per-sample KL term of the doubled batch as one partial sum per frame)."""
    return _EncoderHeadRl.apply(mean, v, w1, b1, w2, b2, fill, u2, eps, mask_bt)


# --------------------------------------------------------------------------------------------- the rl flavour's latent gate
class _RlGate(torch.autograd.Function):
    """Pair doubling + Bernoulli frame masks + fill * (1 - mask) + z * mask of rl_model.VideoVAE (reference train/rl_model.py:136-145) in one
    launch each way (vvae_rl_gate_fwd / _bwd) -> (compressed_representation bf16 (2b, t, hw, ld), selection_mask fp32 (2b, t, 1, 1))."""

    @staticmethod
    def forward(ctx, z, prob, u, fill):
        b, t, hw, ld = z.shape
        z = z.contiguous()
        pr = prob.reshape(b, t).to(torch.float32).contiguous()
        uu = u.reshape(2 * b, t).to(torch.float32).contiguous()
        ff = _f32(fill).reshape(-1)
        comp = torch.empty((2 * b, t, hw, ld), dtype=torch.bfloat16, device=z.device)
        mask = torch.empty((2 * b, t, 1, 1), dtype=torch.float32, device=z.device)
        check(lib().vvae_rl_gate_fwd(_p(z), _p(pr), _p(uu), _p(ff), _p(comp), _p(mask), 2 * b, t, hw * ld, ld, _stream()), "vvae_rl_gate_fwd")
        ctx.save_for_backward(mask)
        ctx.fill = fill
        ctx.zshape = z.shape
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)
        return comp, mask

    @staticmethod
    def backward(ctx, dcomp, _dmask):
        if dcomp is None:
            return None, None, None, None
        (mask,) = ctx.saved_tensors
        b, t, hw, ld = ctx.zshape
        dcomp = dcomp.to(torch.bfloat16).contiguous()
        dz = torch.empty(ctx.zshape, dtype=torch.float32, device=dcomp.device)
        part = torch.empty((lib().vvae_rl_gate_blocks(2 * b, t, hw * ld), ld), dtype=torch.float32, device=dcomp.device)
        check(lib().vvae_rl_gate_bwd(_p(dcomp), _p(mask), _p(dz), _p(part), 2 * b, t, hw * ld, ld, _stream()), "vvae_rl_gate_bwd")
        dfill, _ = fold_partials(part, ctx.fill, None, ld)
        return dz, None, None, (None if dfill is None else dfill.reshape(ctx.fill.shape).to(ctx.fill.dtype))


def rl_gate_ok(z, prob, fill):
    return (z.is_cuda and z.dim() == 4 and z.dtype == torch.float32 and fill.numel() == z.shape[-1] and fill.dtype == torch.float32
            and prob.numel() == z.shape[0] * z.shape[1] and lib().vvae_rl_gate_ok(2 * z.shape[0], z.shape[1], z.shape[2] * z.shape[3], z.shape[3]) == 1)


def rl_gate(z, prob, u, fill):
    """-> (comp bf16 (2b, t, hw, ld), selection_mask fp32 (2b, t, 1, 1)): samples 2k, 2k + 1 share z[k]; mask = u < prob[k]."""
    return _RlGate.apply(z, prob, u, fill)


# --------------------------------------------------------------------------------------------- LayerNorm
def _rows2(x):
    """x (..., C) as rows with two-level strides: row r at (r // inner) * outer_pitch + (r % inner) * inner_pitch.

    Returns (tensor, rows, inner, outer_pitch, inner_pitch); copies only if the layout cannot be expressed that way."""
    c = x.shape[-1]
    n = x.numel() // c
    if x.is_contiguous():
        return x, n, 1, c, 0
    if x.dim() >= 3 and x.stride(-1) == 1:
        inner, ip = x.shape[-2], x.stride(-2)
        lead = x.shape[:-2]
        ok, pitch = True, None
        # leading dims must collapse to one uniform outer pitch
        strides = [x.stride(i) for i in range(len(lead))]
        for i in range(len(lead) - 1):
            if lead[i] != 1 and strides[i] != strides[i + 1] * lead[i + 1]:
                ok = False
        if ok and len(lead) >= 1:
            pitch = strides[-1]
            return x, n, inner, pitch, ip
    x = x.contiguous()
    return x, n, 1, c, 0


class _LayerNorm(torch.autograd.Function):
    """fork=True is the pre-norm residual form ``x + f(LN(x))``: forward also hands x back (an alias), so the gradient of the skip
    path arrives in THIS node's backward and is added inside the LayerNorm-backward kernel instead of by a separate add launch."""

    @staticmethod
    def forward(ctx, x_in, scale, bias, eps, fork):
        x, n, inner, op, ip = _rows2(x_in)
        c = x.shape[-1]
        dt = _dt(x)
        s32 = _f32(scale)
        b32 = _f32(bias) if bias is not None else None
        y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        mean = torch.empty((n,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((n,), dtype=torch.float32, device=x.device)
        check(_launch(f"layernorm_fwd C{c}", 2 * x.numel() * x.element_size(), 0, "layernorm_fwd_kernel",
                      lambda: lib().vvae_layernorm_fwd(_p(x), _p(y), _p(s32), _p(b32), _p(mean), _p(rstd), None, None, n, c, inner, op, ip,
                                                       eps, dt, _stream())), "vvae_layernorm_fwd")
        ctx.save_for_backward(x, s32, mean, rstd)
        ctx.args = (n, c, inner, op, ip, scale.dtype, bias is not None)
        ctx.sparam, ctx.bparam = scale, bias
        ctx.set_materialize_grads(False)
        if fork:
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        dx, dsc, dbi = _ln_backward(ctx, dy, dskip)
        return dx, dsc, dbi, None, None


def _ln_backward(ctx, dy, dskip):
    """Shared by _LayerNorm and _AddLayerNorm: LayerNorm backward with the residual-stream gradient added in the kernel."""
    x, s32, mean, rstd = ctx.saved_tensors
    n, c, inner, op, ip, pdtype, has_bias = ctx.args
    dt = _dt(x)
    if dy is None:                                   # only the skip path was used
        return dskip, None, None
    dy = dy.to(x.dtype).contiguous()
    if dskip is not None:
        dskip = dskip.to(x.dtype).contiguous()
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    nblk = lib().vvae_layernorm_bwd_blocks(n, c, dt)
    part = torch.empty((nblk, 2, c), dtype=torch.float32, device=x.device)
    nstreams = 4 if dskip is not None else 3
    check(_launch(f"layernorm_bwd C{c}" + ("+skip" if dskip is not None else ""), nstreams * x.numel() * x.element_size(), 0,
                  "layernorm_bwd_kernel",
                  lambda: lib().vvae_layernorm_bwd(_p(x), _p(dy), _p(s32), _p(mean), _p(rstd), _p(dskip), _p(dx), _p(part), n, c, inner,
                                                   op, ip, dt, _stream())), "vvae_layernorm_bwd")
    g0, g1 = fold_partials(part, ctx.sparam, ctx.bparam, c)
    if g0 is None:
        return dx, None, None
    return dx, g0.to(pdtype), (g1.to(pdtype) if has_bias else None)


class _AddLayerNorm(torch.autograd.Function):
    """(LayerNorm(skip + o), skip + o) in one pass: the residual add that closes a pre-norm block fused with the LayerNorm that opens
    the next one.  Backward is LayerNorm backward + the gradient of the sum, which goes to both addends unchanged."""

    @staticmethod
    def forward(ctx, skip, o, scale, bias, eps):
        skip = skip.contiguous()
        o = o.to(skip.dtype).reshape(skip.shape).contiguous()
        n, c = skip.numel() // skip.shape[-1], skip.shape[-1]
        dt = _dt(skip)
        s32 = _f32(scale)
        b32 = _f32(bias) if bias is not None else None
        y = torch.empty(skip.shape, dtype=skip.dtype, device=skip.device)
        xs = torch.empty(skip.shape, dtype=skip.dtype, device=skip.device)
        mean = torch.empty((n,), dtype=torch.float32, device=skip.device)
        rstd = torch.empty((n,), dtype=torch.float32, device=skip.device)
        check(_launch(f"add_layernorm_fwd C{c}", 4 * skip.numel() * skip.element_size(), 0, "layernorm_fwd_kernel",
                      lambda: lib().vvae_layernorm_fwd(_p(skip), _p(y), _p(s32), _p(b32), _p(mean), _p(rstd), _p(o), _p(xs), n, c, 1, c, 0,
                                                       eps, dt, _stream())), "vvae_layernorm_fwd")
        ctx.save_for_backward(xs, s32, mean, rstd)
        ctx.args = (n, c, 1, c, 0, scale.dtype, bias is not None)
        ctx.sparam, ctx.bparam = scale, bias
        ctx.set_materialize_grads(False)
        return y, xs

    @staticmethod
    def backward(ctx, dy, dxs=None):
        dx, dsc, dbi = _ln_backward(ctx, dy, dxs)
        return dx, dx, dsc, dbi, None


def layer_norm_supported(x):
    return x.is_cuda and x.dtype in DT and lib().vvae_layernorm_supported(x.shape[-1], DT[x.dtype]) == 1


def layer_norm(x, scale, bias=None, eps=1e-6):
    """nnx.LayerNorm over the last axis, fp32 statistics (reference train/layers.py:17,152,155-156,178)."""
    return _LayerNorm.apply(x, scale, bias, eps, False)


def add_layer_norm_fork(skip, o, scale, bias=None, eps=1e-6):
    """-> (LayerNorm(skip + o), skip + o), one kernel pass (reference train/layers.py:212-221: ``x = x + f(...)`` followed by the
    next block's LayerNorm)."""
    return _AddLayerNorm.apply(skip, o, scale, bias, eps)


def layer_norm_fork(x, scale, bias=None, eps=1e-6):
    """-> (LayerNorm(x), x): the second output is x itself, routed through the node so that in ``x_skip + f(y)`` the skip
    gradient is added inside the LayerNorm-backward kernel (pre-norm residual blocks, reference train/layers.py:212-221)."""
    return _LayerNorm.apply(x, scale, bias, eps, True)


def silu_bf16(x):
    """silu(x) for a contiguous bf16 GPU tensor (no autograd: layers._SiluLinearBf16 owns the backward)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() % 8 == 0 and x.data_ptr() % 16 == 0):
        return F.silu(x)
    y = torch.empty_like(x)
    check(_launch(f"silu {x.numel() >> 20}M", 4 * x.numel(), 0, "silu_bf16_kernel", lambda: lib().vvae_silu_bf16(_p(x), _p(y), x.numel(), _stream())),
          "vvae_silu_bf16")
    return y


# --------------------------------------------------------------------------------------------- Linear + bias + residual (library GEMM)
_LT_WS = {}


def linear_residual_ok(x2, wb, bb, res):
    """bf16 GPU operands the residual-accumulating library product takes (contiguous rows, 16-byte aligned)."""
    ts = (x2, wb, res) if bb is None else (x2, wb, res, bb)
    return (all(t.is_cuda and t.dtype == torch.bfloat16 for t in ts) and x2.dim() == 2 and res.dim() == 2 and x2.stride(1) == 1
            and wb.is_contiguous() and res.stride(1) == 1 and res.shape == (x2.shape[0], wb.shape[1]) and x2.shape[1] == wb.shape[0]
            and x2.shape[1] % 8 == 0 and wb.shape[1] % 8 == 0 and x2.stride(0) % 8 == 0 and res.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and res.data_ptr() % 16 == 0)


def linear_residual(x2, wb, bb, res, wt=None):
    """-> x2 @ wb + bb + res (bf16, fp32 accumulation, one rounding): vvae_linear_residual_bf16.  ``res`` None = plain Linear.
    ``wt``: wb's (N, K) contiguous transpose, if the caller keeps one: the library's faster operand form (vvae_linear_residual_wt_bf16)."""
    m, k = x2.shape
    n = wb.shape[1]
    out = torch.empty((m, n), dtype=torch.bfloat16, device=x2.device)
    ws = _LT_WS.get(x2.device)
    if ws is None:
        ws = _LT_WS[x2.device] = torch.empty(32 << 20, dtype=torch.uint8, device=x2.device)
    ldr = res.stride(0) if res is not None else 0
    if wt is not None and wt.is_contiguous() and wt.shape == (n, k) and wt.dtype == torch.bfloat16:
        fn = lambda: lib().vvae_linear_residual_wt_bf16(_p(x2), x2.stride(0), _p(wt), k, _p(bb), 1, _p(res), ldr, _p(out), n, m, n, k, _p(ws),
                                                        ws.numel(), _stream())
    else:
        fn = lambda: lib().vvae_linear_residual_bf16(_p(x2), x2.stride(0), _p(wb), n, _p(bb), 1, _p(res), ldr, _p(out), n, m, n, k, _p(ws),
                                                     ws.numel(), _stream())
    check(_launch(f"linear+residual {m}x{n} K{k}", (m * k + k * n + (2 if res is not None else 1) * m * n) * 2, 2 * m * n * k, "Cijk_", fn),
          "vvae_linear_residual_bf16")
    return out


# --------------------------------------------------------------------------------------------- dense NT GEMM (Linear fwd / dgrad)
EPI_NONE, EPI_RES, EPI_SILU, EPI_MUL_DSILU = 0, 1, 2, 3


def gemm_nt_supported(a, b):
    return (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
            and a.stride(1) == 1 and b.stride(1) == 1 and a.shape[1] == b.shape[1]
            and lib().vvae_gemm_nt_supported(a.shape[0], b.shape[0], a.shape[1], a.stride(0), b.stride(0), b.shape[0]) == 1)


GEMM_PP = [1]           # 1: the products of gemm_nt run on the second kernel form (csrc/gemm_pp.hip) where it takes the shape; 0: gemm_nt.hip


def gemm_nt(a, b, bias=None, res=None, epi=EPI_NONE, form=None):
    """epi(a (M, K) @ b (N, K)^T + bias) in bf16 with fp32 accumulation (no autograd).  epi = EPI_RES adds ``res`` (M, N);
    EPI_SILU returns (silu(h), h); EPI_MUL_DSILU multiplies by silu'(res).  ``form``: "nt" / "pp" forces a kernel form (tests, A/B)."""
    m, k = a.shape
    n = b.shape[0]
    c = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    c2 = torch.empty((m, n), dtype=torch.bfloat16, device=a.device) if epi == EPI_SILU else None
    if res is not None:
        res = res.reshape(m, n)
        if res.stride(1) != 1:
            res = res.contiguous()
    nbytes = (m * k + n * k + m * n * (1 + (epi != EPI_NONE))) * 2
    use_pp = (form == "pp" or (form is None and GEMM_PP[0])) and lib().vvae_gemm_pp_supported(m, n, k, a.stride(0), b.stride(0), n) == 1
    if form == "pp" and not use_pp:
        raise VvaeError(f"gemm_pp does not take {m} x {n} x {k}")
    if use_pp:
        check(_launch(f"gemm_pp {m}x{n} K{k} epi{epi}", nbytes, 2 * m * n * k, "gemm_pp_kernel",
                      lambda: lib().vvae_gemm_pp_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), n, _p(bias), _p(res),
                                                      res.stride(0) if res is not None else 0, _p(c2), n, epi, m, n, k, _stream())),
              "vvae_gemm_pp_bf16")
        return (c, c2) if epi == EPI_SILU else c
    check(_launch(f"gemm_nt {m}x{n} K{k} epi{epi}", nbytes, 2 * m * n * k, "gemm_nt_kernel",
                  lambda: lib().vvae_gemm_nt_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), n, _p(bias), _p(res),
                                                  res.stride(0) if res is not None else 0, _p(c2), n, epi, m, n, k, _stream())),
          "vvae_gemm_nt_bf16")
    return (c, c2) if epi == EPI_SILU else c


# --------------------------------------------------------------------------------------------- deferred, grouped dense weight gradients
WGRAD_QUEUE = [None]          # a list while a ``deferred_wgrad`` block is open: Linear backward parks (x, dy, kernel, bias) there
GROUP_MAX = 64                # products per grouped launch (kernel-argument table)
GROUP_TILES = 512             # close a group once it holds about this many 256x256 tiles: two full rounds over 256 CUs
GROUP_MIN_TILES = 128         # below this many 256x256 tiles a launch cannot fill the chip without splitting K: per-product path


class deferred_wgrad:
    """``with deferred_wgrad(optimizer): loss.backward()``.

    Weight gradients are not on backward's critical path (only the input gradients are), so the Linear layers of the transformer
    trunk park their (input, output-gradient) pairs instead of launching one split-K product each; whenever about GROUP_TILES
    256x256 tiles have been parked (and on exit) the pairs are multiplied in ONE grouped launch -- hundreds of whole-K tiles at
    once, so nothing is split and nothing is reduced -- straight into the parameters' slots of the optimizer's flat gradient buffer
    (``param.gview``).  The optimizer is told (``optimizer.mark_external``) so that it neither zeroes nor copies those slots and
    can hand completed gradient buckets to the data-parallel reducer while backward is still running.
    """

    def __init__(self, optimizer):
        self.opt = optimizer

    def __enter__(self):
        WGRAD_QUEUE[0] = _WgradQueue(self.opt)
        return self

    def __exit__(self, et, ev, tb):
        q, WGRAD_QUEUE[0] = WGRAD_QUEUE[0], None
        if et is None:
            q.flush()
            q.flush_folds()
        return False


FOLD_MAX = 64                 # partial buffers per grouped fold launch


class _WgradQueue:
    def __init__(self, optimizer):
        self.opt, self.items, self.tiles, self.k, self.seen = optimizer, [], 0, None, set()
        self.folds = []                                      # parked (partial rows, param0, param1) of LayerNorm / q-k-norm scales

    def append_fold(self, part, p0, p1):
        self.folds.append((part, p0, p1))
        if len(self.folds) == FOLD_MAX:
            self.flush_folds()

    @staticmethod
    def _fold_launch(entries):
        """entries: (part pointer, d0 pointer, d1 pointer or None, rows, cols, n0), at most FOLD_MAX per launch."""
        for i0 in range(0, len(entries), FOLD_MAX):
            e = entries[i0:i0 + FOLD_MAX]
            n = len(e)
            VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
            check(lib().vvae_fold_rows_grouped(VP(*[x[0] for x in e]), VP(*[x[1] for x in e]), VP(*[x[2] for x in e]),
                                               IA(*[x[3] for x in e]), IA(*[x[4] for x in e]), IA(*[x[5] for x in e]), n, _stream()),
                  "vvae_fold_rows_grouped")

    def flush_folds(self):
        folds, self.folds = self.folds, []
        if not folds:
            return
        first, second, keep = [], [], []
        for part, p0, p1 in folds:
            rows = part.shape[0]
            cols = part.numel() // rows
            d0, d1 = p0.gview.data_ptr(), (p1.gview.data_ptr() if p1 is not None else None)
            r, c, k = rows, cols, 1
            while r > 1024 and r % 2 == 0 and c * 2 <= 8192:          # tall and narrow (one row per attention workgroup): read
                r //= 2; c *= 2; k *= 2                               # (rows, cols) as (rows/k, k*cols) so that many workgroups
            if k == 1:                                                # share the fold, then sum the k partial rows
                first.append((part.data_ptr(), d0, d1, rows, cols, p0.numel()))
            else:
                tmp = torch.empty((k, cols), dtype=torch.float32, device=part.device)
                keep.append(tmp)
                first.append((part.data_ptr(), tmp.data_ptr(), None, r, c, c))
                second.append((tmp.data_ptr(), d0, d1, k, cols, p0.numel()))
        self._fold_launch(first)
        self._fold_launch(second)
        for _, p0, p1 in folds:
            self.opt.mark_external(p0)
            if p1 is not None:
                self.opt.mark_external(p1)

    def claim(self, kernel):
        if id(kernel) in self.seen:                          # a weight used twice in one step would need accumulation
            raise VvaeError("deferred_wgrad: a kernel was used twice in one backward pass")
        self.seen.add(id(kernel))

    def append(self, item):
        x2, dy2, kernel, bias = item
        self.claim(kernel)
        t = (x2.shape[1] // 256) * (dy2.shape[1] // 256)
        if self.items and (x2.shape[0] != self.k or len(self.items) == GROUP_MAX or self.tiles + t > GROUP_TILES):
            self.flush()
        self.items.append(item); self.tiles += t; self.k = x2.shape[0]

    def flush(self):
        items, tiles, k = self.items, self.tiles, self.k
        self.items, self.tiles = [], 0
        if items:
            flush_wgrad(items, self.opt, tiles, k)


def wgrad_deferrable(x2, dy2, kernel, bias):
    """Linear.backward asks: may this product be parked?  (bf16 GPU operands, flat-buffer slots present, 256-multiples.)"""
    if WGRAD_QUEUE[0] is None or getattr(kernel, "gview", None) is None or (bias is not None and getattr(bias, "gview", None) is None):
        return False
    k, m = x2.shape
    n = dy2.shape[1]
    return (x2.is_cuda and x2.dtype == torch.bfloat16 and dy2.dtype == torch.bfloat16 and x2.stride(1) == 1 and dy2.stride(1) == 1
            and m % 256 == 0 and n % 256 == 0 and k % 32 == 0 and x2.stride(0) % 8 == 0 and dy2.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and dy2.data_ptr() % 16 == 0)


def flush_wgrad(items, optimizer, tiles=None, k=None):
    """Multiply the parked pairs ``(x (K, M), dy (K, N), kernel, bias)`` (all with the same K, at most GROUP_MAX of them)."""
    if not items:
        return
    if tiles is None:
        tiles = sum((x2.shape[1] // 256) * (dy2.shape[1] // 256) for x2, dy2, _, _ in items)
        k = items[0][0].shape[0]
    if os.environ.get("VVAE_WGRAD_DEBUG"):
        print(f"[vvae] dense dW flush: {len(items)} products, {tiles} tiles of 256 x 256, K {k}: " + ("grouped" if tiles >= GROUP_MIN_TILES else "per product"), file=sys.stderr)
    if tiles >= GROUP_MIN_TILES:
        _gemm_tn_grouped(items, k)
    else:                                                    # too few tiles to fill the chip without splitting K
        for x2, dy2, kernel, bias in items:                  # straight into the flat-buffer slots: no copies behind the product
            gemm_tn(x2, dy2, bias is not None, kernel.gview, bias.gview if bias is not None else None)
    for _, _, kernel, bias in items:
        optimizer.mark_external(kernel)
        if bias is not None:
            optimizer.mark_external(bias)


def _gemm_tn_grouped(chunk, k):
    n = len(chunk)
    VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
    a = VP(*[x2.data_ptr() for x2, _, _, _ in chunk])
    b = VP(*[dy2.data_ptr() for _, dy2, _, _ in chunk])
    c = VP(*[kn.gview.data_ptr() for _, _, kn, _ in chunk])
    db = VP(*[(bi.gview.data_ptr() if bi is not None else None) for _, _, _, bi in chunk])
    lda = IA(*[x2.stride(0) for x2, _, _, _ in chunk])
    ldb = IA(*[dy2.stride(0) for _, dy2, _, _ in chunk])
    ms = IA(*[x2.shape[1] for x2, _, _, _ in chunk])
    ns = IA(*[dy2.shape[1] for _, dy2, _, _ in chunk])
    flops = sum(2 * k * x2.shape[1] * dy2.shape[1] for x2, dy2, _, _ in chunk)
    nbytes = sum(k * (x2.shape[1] + dy2.shape[1]) * 2 + x2.shape[1] * dy2.shape[1] * 4 for x2, dy2, _, _ in chunk)
    check(_launch(f"gemm_tn_grouped x{n} K{k}", nbytes, flops, "gemm_tn256_grouped_kernel",
                  lambda: lib().vvae_gemm_tn_grouped_bf16(a, lda, b, ldb, c, db, ms, ns, n, k, _stream())), "vvae_gemm_tn_grouped_bf16")


# --------------------------------------------------------------------------------------------- dense weight-gradient GEMM
def gemm_tn_supported(a, b):
    return (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
            and a.stride(1) == 1 and b.stride(1) == 1
            and lib().vvae_gemm_tn_supported(a.shape[1], b.shape[1], a.shape[0], a.stride(0), b.stride(0)) == 1)


def gemm_tn(a, b, want_colsum=True, out=None, out_colsum=None):
    """(a^T @ b, b.sum(0)) in fp32 for bf16 token-major a (K, M), b (K, N): dW and db of a Linear layer in one pass.
    ``out`` / ``out_colsum``: contiguous fp32 destinations (the parameters' slots of the flat gradient buffer) instead of fresh tensors."""
    k, m = a.shape
    n = b.shape[1]
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    db = (out_colsum if out_colsum is not None else torch.empty((n,), dtype=torch.float32, device=a.device)) if want_colsum else None
    assert c.is_contiguous() and c.dtype == torch.float32 and c.shape == (m, n)
    wsb = lib().vvae_gemm_tn_ws_bytes(m, n, k)
    ws, wsb = _ws(wsb, a.device)
    kern = ("gemm_tn256_kernel" if m % 256 == 0 and n % 256 == 0 and k % 32 == 0 else "gemm_tn_bf16_kernel") + " + gemm_tn_reduce_kernel"
    check(_launch(f"gemm_tn {m}x{n} K{k}", (k * (m + n)) * 2 + m * n * 4, 2 * m * n * k, kern,
                  lambda: lib().vvae_gemm_tn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), _p(db), m, n, k, _p(ws), wsb, _stream())),
          "vvae_gemm_tn_bf16")
    return c, db
