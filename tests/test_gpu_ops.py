"""GPU parity: every HIP op (through the C ABI via video_vae_amd.ops) against the CPU oracle, same seeded inputs.

fp32 bar: rtol 1e-3 / atol 1e-4 (BASELINE.json:north_star).  bf16 storage: the oracle is run with dtype=bf16
emulation and compared at bf16 resolution (rtol 2e-2, atol scaled).
"""
import math
import pytest
import torch

from oracle import nn as O
from oracle import layers as OL
from oracle import loss as OLoss
from util import assert_close, assert_close_scaled, rnd

pytestmark = pytest.mark.gpu


def _ops():
    from video_vae_amd import ops
    return ops


CONV_CASES = [
    # n, t, h, w, cin, cout, kt, kh, kw
    (1, 4, 8, 8, 5, 7, 3, 3, 3),
    (2, 3, 6, 10, 12, 12, 3, 7, 7),
    (1, 2, 4, 4, 16, 3, 1, 1, 1),
    (1, 3, 9, 7, 16, 16, 3, 3, 3),
    (2, 2, 8, 8, 32, 16, 3, 3, 3),
    (1, 1, 5, 5, 4, 20, 1, 3, 3),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3d_fwd_bwd(dev, case, dtype):
    ops = _ops()
    n, t, h, w, ci, co, kt, kh, kw = case
    x = rnd((n, t, h, w, ci), 1)
    k = rnd((kt, kh, kw, ci, co), 2, (kt * kh * kw * ci) ** -0.5)
    b = rnd((co,), 3, 0.1)
    gy = rnd((n, t, h, w, co), 4)
    od = dtype
    xq = x.to(dtype).float()
    xo = xq.clone().requires_grad_(True); ko = k.clone().requires_grad_(True); bo = b.clone().requires_grad_(True)
    yo = O.conv3d_same(xo, ko, bo, od)
    yo.backward(gy.to(dtype).float())
    xg = xq.to(dev, dtype).requires_grad_(True); kg = k.to(dev).requires_grad_(True); bg = b.to(dev).requires_grad_(True)
    yg = ops.conv3d(xg, kg, bg)
    yg.backward(gy.to(dev, dtype))
    if dtype == torch.float32:
        assert_close(yg, yo, what="y")
        assert_close_scaled(xg.grad, xo.grad, what="dx")
        assert_close_scaled(kg.grad, ko.grad, what="dw")
        assert_close_scaled(bg.grad, bo.grad, what="db")
    else:
        assert_close(yg, yo, rtol=2e-2, atol=2e-2, what="y")
        assert_close_scaled(xg.grad, xo.grad, rel=2e-2, what="dx")
        assert_close_scaled(kg.grad, ko.grad, rel=2e-2, what="dw")
        assert_close_scaled(bg.grad, bo.grad, rel=2e-2, what="db")


def test_conv3d_channel_slice_operands(dev):
    """Operands that are channel slices of wider buffers (row pitch > C): the concat-elision case."""
    ops = _ops()
    x = rnd((1, 2, 6, 6, 24), 5)
    k = rnd((3, 3, 3, 8, 16), 6, 0.1)
    xs = x[..., 8:16]
    want = O.conv3d_same(xs, k, None)
    xg = x.to(dev)
    out = torch.zeros((1, 2, 6, 6, 40), device=dev)
    got = ops.conv3d_fwd_raw(xg[..., 8:16], k.to(dev), None, out=out[..., 16:32])
    assert got.data_ptr() == out[..., 16:32].data_ptr()
    assert_close(out[..., 16:32], want, what="sliced y")
    assert float(out[..., :16].abs().max()) == 0 and float(out[..., 32:].abs().max()) == 0


@pytest.mark.parametrize("shape,groups", [((2, 3, 6, 6, 16), 8), ((1, 2, 4, 4, 12), 4), ((2, 2, 4, 6, 128), 8), ((1, 4, 8, 8, 3), 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_group_norm_silu(dev, shape, groups, dtype):
    ops = _ops()
    c = shape[-1]
    x = (rnd(shape, 7) * 1.5 + 0.3).to(dtype).float()
    sc = 1 + 0.2 * rnd((c,), 8); bi = 0.1 * rnd((c,), 9)
    gy = rnd(shape, 10).to(dtype).float()
    xo = x.clone().requires_grad_(True); so = sc.clone().requires_grad_(True); bo = bi.clone().requires_grad_(True)
    yo = O.silu(O.group_norm(xo, so, bo, groups, dtype), dtype)
    yo.backward(gy)
    xg = x.to(dev, dtype).requires_grad_(True); sg = sc.to(dev).requires_grad_(True); bg = bi.to(dev).requires_grad_(True)
    yg = ops.group_norm_silu(xg, sg, bg, groups, 1e-6)
    yg.backward(gy.to(dev, dtype))
    if dtype == torch.float32:
        assert_close(yg, yo, what="y")
        assert_close_scaled(xg.grad, xo.grad, what="dx")
        assert_close_scaled(sg.grad, so.grad, what="dscale")
        assert_close_scaled(bg.grad, bo.grad, what="dbias")
    else:
        assert_close(yg, yo, rtol=2e-2, atol=2e-2, what="y")
        assert_close_scaled(xg.grad, xo.grad, rel=3e-2, what="dx")
        assert_close_scaled(sg.grad, so.grad, rel=3e-2, what="dscale")
        assert_close_scaled(bg.grad, bo.grad, rel=3e-2, what="dbias")


def test_group_norm_stats_span_time(dev):
    """GroupNorm statistics span (t,h,w): changing frame k must change the normalised output of frame j != k."""
    ops = _ops()
    x = rnd((1, 4, 4, 4, 8), 11).to(dev)
    sc = torch.ones(8, device=dev); bi = torch.zeros(8, device=dev)
    y0 = ops.group_norm_silu(x, sc, bi, 8)
    x2 = x.clone(); x2[:, 3] += 5.0
    y1 = ops.group_norm_silu(x2, sc, bi, 8)
    assert float((y0[:, 0] - y1[:, 0]).abs().max()) > 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c", [16, 6])
def test_max_pool(dev, dtype, c):
    ops = _ops()
    x = rnd((2, 3, 8, 6, c), 12).to(dtype).float()
    gy = rnd((2, 3, 4, 3, c), 13).to(dtype).float()
    xo = x.clone().requires_grad_(True)
    yo = O.max_pool_1x2x2(xo)
    yo.backward(gy)
    xg = x.to(dev, dtype).requires_grad_(True)
    yg = ops.max_pool_1x2x2(xg)
    yg.backward(gy.to(dev, dtype))
    assert torch.equal(yg.float().cpu(), yo.detach())
    assert torch.equal(xg.grad.float().cpu(), xo.grad)


def test_max_pool_ties_first_wins(dev):
    ops = _ops()
    x = torch.ones((1, 1, 2, 2, 8), device=dev, requires_grad=True)
    ops.max_pool_1x2x2(x).sum().backward()
    want = torch.zeros(1, 1, 2, 2, 8); want[0, 0, 0, 0] = 1
    assert torch.equal(x.grad.cpu(), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ci,co", [(32, 16), (5, 9), (128, 64)])
def test_conv_transpose(dev, dtype, ci, co):
    ops = _ops()
    x = rnd((2, 2, 3, 5, ci), 14).to(dtype).float()
    k = rnd((1, 2, 2, ci, co), 15, (4 * ci) ** -0.5); b = rnd((co,), 16, 0.1)
    gy = rnd((2, 2, 6, 10, co), 17).to(dtype).float()
    xo = x.clone().requires_grad_(True); ko = k.clone().requires_grad_(True); bo = b.clone().requires_grad_(True)
    yo = O.conv_transpose_1x2x2(xo, ko, bo, dtype)
    yo.backward(gy)
    xg = x.to(dev, dtype).requires_grad_(True); kg = k.to(dev).requires_grad_(True); bg = b.to(dev).requires_grad_(True)
    yg = ops.conv_transpose_1x2x2(xg, kg, bg)
    yg.backward(gy.to(dev, dtype))
    r = 1e-3 if dtype == torch.float32 else 2e-2
    assert_close(yg, yo, rtol=r, atol=(1e-4 if dtype == torch.float32 else 2e-2), what="y")
    assert_close_scaled(xg.grad, xo.grad, rel=r, what="dx")
    assert_close_scaled(kg.grad, ko.grad, rel=r, what="dw")
    assert_close_scaled(bg.grad, bo.grad, rel=r, what="db")


def test_conv_transpose_disjoint_blocks(dev):
    """Each input voxel maps to a disjoint 2x2 output block: a one-hot input lights exactly 4 output voxels."""
    ops = _ops()
    x = torch.zeros((1, 1, 3, 3, 4), device=dev); x[0, 0, 1, 2, :] = 1
    k = rnd((1, 2, 2, 4, 4), 18).to(dev)
    y = ops.conv_transpose_1x2x2(x, k, torch.zeros(4, device=dev))
    nz = (y.abs().sum(-1) > 0).nonzero().cpu().tolist()
    assert sorted(map(tuple, nz)) == [(0, 0, 2, 4), (0, 0, 2, 5), (0, 0, 3, 4), (0, 0, 3, 5)]


def _attn_ref(qkv, qs, ks, mask, heads, max_len, dtype):
    a, t, c3 = qkv.shape
    q, k, v = torch.chunk(qkv, 3, dim=-1)
    sp = lambda z: z.reshape(a, t, heads, -1)
    q, k, v = sp(q), sp(k), sp(v)
    q = O.layer_norm(q, qs, None, dtype); k = O.layer_norm(k, ks, None, dtype)
    cos, sin = OL.rope_tables(q.shape[-1], max_len)
    q, k = OL.rope(q, k, cos, sin, dtype)
    return OL.dot_product_attention(q, k, v, mask, dtype).reshape(a, t, -1)


@pytest.mark.parametrize("a,t,heads,d", [(6, 16, 8, 64), (5, 5, 4, 32), (3, 32, 2, 64), (2, 64, 1, 64), (4, 7, 2, 16),
                                         (9, 16, 4, 8), (3, 12, 2, 24), (2, 40, 2, 32), (3, 20, 3, 64)])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("generic", [False, True])
def test_temporal_attention_core(dev, a, t, heads, d, masked, dtype, generic):
    """generic=False: lane-per-frame kernels (head_dim 8/16/32/64); generic=True (and head_dim 24): the any-shape kernels."""
    ops = _ops()
    ops.ATTN_FORCE_GENERIC[0] = generic
    qkv = rnd((a, t, 3 * heads * d), 19).to(dtype).float()
    qs = 1 + 0.2 * rnd((d,), 20); ks = 1 + 0.2 * rnd((d,), 21)
    go = rnd((a, t, heads * d), 22).to(dtype).float()
    mask = None
    if masked:
        lens = torch.tensor([max(1, t - (i * 3) % t) for i in range(a)])
        mask = (torch.arange(t)[None, :] < lens[:, None]).reshape(a, 1, 1, t)
    cos, sin = OL.rope_tables(d, 64)
    xo = qkv.clone().requires_grad_(True); qso = qs.clone().requires_grad_(True); kso = ks.clone().requires_grad_(True)
    yo = _attn_ref(xo, qso, kso, mask, heads, 64, dtype)
    yo.backward(go)
    xg = qkv.to(dev, dtype).requires_grad_(True); qsg = qs.to(dev).requires_grad_(True); ksg = ks.to(dev).requires_grad_(True)
    m8 = mask.reshape(a, t).to(torch.uint8).to(dev) if masked else None
    try:
        yg = ops.temporal_attention_core(xg, qsg, ksg, cos.to(dev), sin.to(dev), m8, 1, heads)
        yg.backward(go.to(dev, dtype))
    finally:
        ops.ATTN_FORCE_GENERIC[0] = False
    if dtype == torch.float32:
        assert_close(yg, yo, what="out")
        assert_close_scaled(xg.grad, xo.grad, what="dqkv")
        assert_close_scaled(qsg.grad, qso.grad, what="dq_scale")
        assert_close_scaled(ksg.grad, kso.grad, what="dk_scale")
    else:
        assert_close(yg, yo, rtol=3e-2, atol=3e-2, what="out")
        assert_close_scaled(xg.grad, xo.grad, rel=5e-2, what="dqkv")
        assert_close_scaled(qsg.grad, qso.grad, rel=5e-2, what="dq_scale (bf16)")
        assert_close_scaled(ksg.grad, kso.grad, rel=5e-2, what="dk_scale (bf16)")


@pytest.mark.parametrize("a,s,heads,d", [(3, 256, 8, 64), (2, 100, 4, 32), (2, 40, 3, 16), (1, 70, 2, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_qk_prep_kernels(dev, a, s, heads, d, dtype):
    """q/k-norm + RoPE prep pass (fwd and bwd, v-gradient copy included) against the oracle's LayerNorm + RoPE."""
    ops = _ops()
    hd = heads * d
    qkv = rnd((a, s, 3 * hd), 60).to(dtype).float()
    qs = 1 + 0.2 * rnd((d,), 61); ks = 1 + 0.2 * rnd((d,), 62)
    gq = rnd((a, heads, s, d), 63).to(dtype).float(); gk = rnd((a, heads, s, d), 64).to(dtype).float()
    gv = rnd((a, heads, s, d), 65).to(dtype).float()
    cos, sin = OL.rope_tables(d, 256)
    xo = qkv.clone().requires_grad_(True); qso = qs.clone().requires_grad_(True); kso = ks.clone().requires_grad_(True)
    q, k, v = torch.chunk(xo, 3, dim=-1)
    sp = lambda z: z.reshape(a, s, heads, d)
    qn = O.layer_norm(sp(q), qso, None, dtype); kn = O.layer_norm(sp(k), kso, None, dtype)
    qr, kr = OL.rope(qn, kn, cos, sin, dtype)
    ((qr.transpose(1, 2) * gq).sum() + (kr.transpose(1, 2) * gk).sum() + (sp(v).transpose(1, 2) * gv).sum()).backward()
    xg = qkv.to(dev, dtype)
    qk = ops.qk_prep_fwd_raw(xg, qs.to(dev), ks.to(dev), cos.to(dev), sin.to(dev), heads)
    # gradients handed over in a non-trivial layout: (a, s, heads, d) storage viewed as (a, heads, s, d)
    lay = lambda g: g.to(dev, dtype).transpose(1, 2).contiguous().transpose(1, 2)
    dqkv, dqs, dks = ops.qk_prep_bwd_raw(xg, lay(gq), gk.to(dev, dtype), lay(gv), qs.to(dev), ks.to(dev), cos.to(dev), sin.to(dev), heads)
    ref_qk = torch.cat([qr.reshape(a, s, hd), kr.reshape(a, s, hd)], -1)
    if dtype == torch.float32:
        assert_close(qk, ref_qk, what="qk")
        assert_close_scaled(dqkv, xo.grad, what="dqkv")
        assert_close_scaled(dqs, qso.grad, what="dq_scale")
        assert_close_scaled(dks, kso.grad, what="dk_scale")
    else:
        assert_close(qk, ref_qk, rtol=2e-2, atol=2e-2, what="qk")
        assert_close_scaled(dqkv, xo.grad, rel=3e-2, what="dqkv")
        assert_close_scaled(dqs, qso.grad, rel=3e-2, what="dq_scale")
        assert_close_scaled(dks, kso.grad, rel=3e-2, what="dk_scale")
    assert torch.equal(dqkv[..., 2 * hd:].float().cpu(), gv.transpose(1, 2).reshape(a, s, hd).to(dtype).float()), "dv is a copy"


@pytest.mark.parametrize("a,s,heads,d", [(3, 256, 8, 64), (2, 96, 4, 32), (2, 160, 3, 64), (1, 32, 2, 64), (2, 224, 2, 64)])
@pytest.mark.parametrize("library_core", [False, True])
def test_spatial_attention_core_bf16(dev, a, s, heads, d, library_core):
    """Spatial attention against the oracle (reference train/layers.py:153-170), forward and backward.
    library_core=False: the fused MFMA kernels where they apply (head_dim 64); True: prep kernels + library flash core."""
    ops = _ops()
    ops.SPATIAL_FORCE_LIBRARY_CORE[0] = library_core
    try:
        _spatial_core_case(ops, dev, a, s, heads, d)
    finally:
        ops.SPATIAL_FORCE_LIBRARY_CORE[0] = False


def _spatial_core_case(ops, dev, a, s, heads, d):
    dtype = torch.bfloat16
    qkv = rnd((a, s, 3 * heads * d), 70).to(dtype).float()
    qs = 1 + 0.2 * rnd((d,), 71); ks = 1 + 0.2 * rnd((d,), 72)
    go = rnd((a, s, heads * d), 73).to(dtype).float()
    cos, sin = OL.rope_tables(d, 256)
    xo = qkv.clone().requires_grad_(True); qso = qs.clone().requires_grad_(True); kso = ks.clone().requires_grad_(True)
    yo = _attn_ref(xo, qso, kso, None, heads, 256, dtype)
    yo.backward(go)
    xg = qkv.to(dev, dtype).requires_grad_(True); qsg = qs.to(dev).requires_grad_(True); ksg = ks.to(dev).requires_grad_(True)
    assert ops.spatial_attention_supported(xg, heads, 256)
    yg = ops.spatial_attention_core(xg, qsg, ksg, cos.to(dev), sin.to(dev), heads)
    yg.backward(go.to(dev, dtype))
    assert_close(yg, yo, rtol=3e-2, atol=3e-2, what="out")
    assert_close_scaled(xg.grad, xo.grad, rel=5e-2, what="dqkv")
    assert_close_scaled(qsg.grad, qso.grad, rel=5e-2, what="dq_scale")
    assert_close_scaled(ksg.grad, kso.grad, rel=5e-2, what="dk_scale")


@pytest.mark.parametrize("a,s,heads", [(3, 256, 8), (2, 96, 4), (5, 32, 2), (1, 160, 3)])
def test_spatial_attention_fused_forward(dev, a, s, heads):
    """One-kernel q/k-norm + RoPE + softmax(QK^T/sqrt(D))V (MFMA, head_dim 64) against the oracle's attention."""
    ops = _ops()
    d, dtype = 64, torch.bfloat16
    qkv = rnd((a, s, 3 * heads * d), 75).to(dtype).float()
    qs = 1 + 0.2 * rnd((d,), 76); ks = 1 + 0.2 * rnd((d,), 77)
    cos, sin = OL.rope_tables(d, 256)
    yo = _attn_ref(qkv, qs, ks, None, heads, 256, dtype)
    xg = qkv.to(dev, dtype)
    assert ops.spatial_attn_fused_supported(xg, heads)
    yg, lse2 = ops.spatial_attn_fwd_raw(xg, qs.to(dev), ks.to(dev), cos.to(dev), sin.to(dev), heads)
    assert_close(yg, yo, rtol=3e-2, atol=3e-2, what="out")
    # log-sum-exp against an fp32 recomputation from the oracle's rotated q, k
    q, k, v = torch.chunk(qkv, 3, dim=-1)
    sp = lambda z: z.reshape(a, s, heads, d)
    qr, kr = OL.rope(O.layer_norm(sp(q), qs, None, dtype), O.layer_norm(sp(k), ks, None, dtype), cos, sin, dtype)
    sc = torch.einsum("aqhd,akhd->ahqk", qr, kr) / d ** 0.5
    want = torch.logsumexp(sc, -1).reshape(a * heads, s) / math.log(2.0)
    assert_close(lse2, want, rtol=1e-2, atol=2e-2, what="lse2")


def test_temporal_attention_masked_equals_truncated(dev):
    """Reference property (train/scratch.py:46-57): keys >= L masked  ==  attention over the first L frames."""
    ops = _ops()
    a, t, heads, d, L = 4, 32, 2, 64, 10
    qkv = rnd((a, t, 3 * heads * d), 23).to(dev)
    qs = torch.ones(d, device=dev); ks = torch.ones(d, device=dev)
    cos, sin = OL.rope_tables(d, 64)
    cos, sin = cos.to(dev), sin.to(dev)
    m8 = (torch.arange(t) < L).to(torch.uint8)[None].repeat(a, 1).to(dev)
    full = ops.temporal_attention_core(qkv, qs, ks, cos, sin, m8, 1, heads)
    trunc = ops.temporal_attention_core(qkv[:, :L].contiguous(), qs, ks, cos, sin, None, 1, heads)
    assert_close(full[:, :L], trunc, what="masked vs truncated")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reparam_kl(dev, dtype):
    ops = _ops()
    b, t, hw, c = 3, 6, 4, 24
    mean = rnd((b, t, hw, c), 24).to(dtype).float(); lv = (0.5 * rnd((b, t, hw, c), 25) - 1).to(dtype).float()
    eps = rnd((b, t, hw, c), 26)
    mask = torch.ones(b, t); mask[1, 4:] = 0; mask[2, 1:] = 0
    gz = rnd((b, t, hw, c), 27); gk = rnd((b,), 28)
    mo = mean.clone().requires_grad_(True); lo = lv.clone().requires_grad_(True)
    zo = mo + eps * torch.exp(lo / 2)
    klo = OLoss.kl_per_sample(mo, lo, mask)
    (zo * gz).sum().add((klo * gk).sum()).backward()
    mg = mean.to(dev, dtype).requires_grad_(True); lg = lv.to(dev, dtype).requires_grad_(True)
    zg, klg = ops.reparameterise_kl(mg, lg, eps.to(dev), mask.to(dev))
    ((zg * gz.to(dev)).sum() + (klg * gk.to(dev)).sum()).backward()
    r = 1e-3 if dtype == torch.float32 else 2e-2
    assert_close(zg, zo, rtol=r, atol=1e-4, what="z")
    assert_close(klg, klo, rtol=r, atol=1e-5, what="kl")
    assert_close_scaled(mg.grad, mo.grad, rel=r, what="dmean")
    assert_close_scaled(lg.grad, lo.grad, rel=r, what="dlogvar")
    # the two single-output forms
    z2 = ops.reparameterise(mg.detach(), lg.detach(), eps.to(dev))
    k2 = ops.kl_per_sample(mg.detach(), lg.detach(), mask.to(dev))
    assert_close(z2, zo, rtol=r, atol=1e-4); assert_close(k2, klo, rtol=r, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("div", [1, 2])
def test_masked_mse_mae(dev, dtype, div):
    ops = _ops()
    b, t, h, w, c = 4, 5, 6, 6, 3
    video = torch.rand((b // div, t, h, w, c), generator=torch.Generator().manual_seed(29)).to(dtype).float()
    recon = (video.repeat_interleave(div, 0) + 0.3 * rnd((b, t, h, w, c), 30)).to(dtype).float()
    mask = torch.ones(b, t); mask[1, 3:] = 0; mask[3, 1:] = 0
    g2 = rnd((b,), 31); g1 = rnd((b,), 32)
    ro = recon.clone().requires_grad_(True)
    mse_o, mae_o = OLoss.masked_mse_mae(video.repeat_interleave(div, 0), ro, mask)
    ((mse_o * g2).sum() + (mae_o * g1).sum()).backward()
    rg = recon.to(dev, dtype).requires_grad_(True)
    mse_g, mae_g = ops.masked_mse_mae(video.to(dev, dtype), rg, mask.to(dev), div)
    ((mse_g * g2.to(dev)).sum() + (mae_g * g1.to(dev)).sum()).backward()
    r = 1e-3 if dtype == torch.float32 else 2e-2
    assert_close(mse_g, mse_o, rtol=r, atol=1e-6, what="mse")
    assert_close(mae_g, mae_o, rtol=r, atol=1e-6, what="mae")
    assert_close_scaled(rg.grad, ro.grad, rel=r, what="drecon")


# ------------------------------------------------------------------------------------------- bf16 MFMA fast path
FAST_CASES = [
    # cin, cout, kt, kh, kw, (n, t, h, w)
    (16, 16, 3, 3, 3, (1, 3, 20, 24)), (16, 32, 3, 3, 3, (2, 2, 9, 17)), (32, 32, 3, 3, 3, (1, 4, 16, 16)),
    (32, 64, 3, 3, 3, (1, 2, 12, 20)), (64, 64, 3, 3, 3, (1, 3, 8, 16)), (64, 128, 3, 3, 3, (1, 2, 8, 16)),
    (128, 128, 3, 3, 3, (1, 2, 8, 8)), (128, 64, 3, 3, 3, (1, 2, 10, 16)), (64, 32, 3, 3, 3, (1, 2, 16, 16)),
    (32, 16, 3, 3, 3, (1, 3, 16, 32)), (16, 16, 3, 7, 7, (1, 3, 20, 24)), (16, 16, 3, 7, 7, (2, 1, 5, 40)),
    (32, 16, 3, 3, 3, (1, 2, 3, 50)), (32, 32, 3, 3, 3, (2, 1, 1, 1)), (16, 16, 3, 7, 7, (1, 2, 2, 3)),      # degenerate extents
]


def _bf16_exact(shape, seed, scale):
    return (rnd(shape, seed, scale)).to(torch.bfloat16).float()


ROLL_CASES = [
    # cin, cout, kh, (n, t, h, w): every single-chunk shape the rolling time-column kernel owns, fwd and dgrad roles
    (16, 16, 3, (2, 5, 20, 24)), (16, 32, 3, (1, 7, 9, 33)), (32, 16, 3, (1, 6, 17, 16)), (32, 32, 3, (2, 5, 18, 30)),
    (16, 16, 7, (1, 5, 21, 40)),
    # degenerate extents: one frame, tiles wider / taller than the volume, a single row
    (16, 16, 3, (1, 1, 5, 7)), (32, 32, 3, (1, 2, 3, 50)), (32, 16, 3, (3, 1, 1, 1)), (16, 16, 7, (1, 2, 2, 3)),
]


@pytest.mark.parametrize("case", ROLL_CASES)
@pytest.mark.parametrize("tchunk", [0, 1, 3, 16])
def test_conv3d_rolling_kernel_matches_per_frame(dev, case, tchunk):
    """The rolling time-column kernel multiplies the same fragments in the same order as the per-frame kernel: outputs must be
    bitwise equal for any frames-per-workgroup split, ragged tiles included."""
    from video_vae_amd import ops
    from video_vae_amd._lib import lib
    ci, co, kh, (n, t, h, w) = case
    x = rnd((n, t, h, w, ci), 60, 1.0).to(dev, torch.bfloat16)
    gy = rnd((n, t, h, w, co), 61, 1.0).to(dev, torch.bfloat16)
    k = rnd((3, kh, kh, ci, co), 62, (3 * kh * kh * ci) ** -0.5).to(dev)
    b = rnd((co,), 63, 0.1).to(dev)
    try:
        lib().vvae_conv3d_roll_config(0, 0)
        y0, dx0 = ops.conv3d_fwd_raw(x, k, b), ops.conv3d_dgrad_raw(gy, k)
        lib().vvae_conv3d_roll_config(1, tchunk)
        y1, dx1 = ops.conv3d_fwd_raw(x, k, b), ops.conv3d_dgrad_raw(gy, k)
    finally:
        lib().vvae_conv3d_roll_config(1, 0)
    assert torch.equal(y0, y1)
    assert torch.equal(dx0, dx1)


DEEP_CASES = [
    # cin, cout, (n, t, h, w): the layers the deep rolling kernel owns (K channels in 1 / 2 / 4 chunks of 32 split over the waves), fwd and
    # dgrad roles, ragged tiles, one frame, volumes smaller than a tile
    (32, 64, (1, 5, 20, 24)), (64, 32, (2, 3, 9, 33)), (64, 64, (1, 6, 17, 16)), (64, 128, (1, 4, 10, 18)), (128, 64, (1, 3, 7, 30)),
    (128, 128, (2, 2, 5, 16)), (64, 64, (1, 1, 3, 5)), (128, 128, (1, 7, 1, 1)), (32, 128, (1, 2, 16, 16)),
]


@pytest.mark.parametrize("case", DEEP_CASES)
def test_conv3d_deep_kernel_matches_per_frame(dev, case):
    """The deep rolling kernel (waves split the K chunks, partial sums folded through LDS in chunk order) against the per-frame kernel it
    replaces: the same products; with one K chunk the same summation order (bitwise), with 2 / 4 chunks the fp32 partial sums are
    associated differently, so a small fraction of the bf16 outputs may sit one rounding step apart.  Deterministic run to run."""
    from video_vae_amd import ops
    from video_vae_amd._lib import lib
    ci, co, (n, t, h, w) = case
    x = rnd((n, t, h, w, ci), 70, 1.0).to(dev, torch.bfloat16)
    gy = rnd((n, t, h, w, co), 71, 1.0).to(dev, torch.bfloat16)
    k = rnd((3, 3, 3, ci, co), 72, (27 * ci) ** -0.5).to(dev)
    b = rnd((co,), 73, 0.1).to(dev)
    try:
        lib().vvae_conv3d_deep_config(0)
        y0, dx0 = ops.conv3d_fwd_raw(x, k, b), ops.conv3d_dgrad_raw(gy, k)
        lib().vvae_conv3d_deep_config(1)
        y1, dx1 = ops.conv3d_fwd_raw(x, k, b), ops.conv3d_dgrad_raw(gy, k)
        y2, dx2 = ops.conv3d_fwd_raw(x, k, b), ops.conv3d_dgrad_raw(gy, k)
    finally:
        lib().vvae_conv3d_deep_config(1)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2), "not reproducible"
    for name, a, ref, ck in (("y", y1, y0, ci), ("dx", dx1, dx0, co)):
        if ck == 32:
            assert torch.equal(a, ref), f"{name}: one K chunk must be bitwise equal to the per-frame kernel"
            continue
        a, ref = a.float(), ref.float()
        frac = float((a != ref).float().mean())
        err = float(((a - ref).abs() / ref.abs().clamp_min(0.25)).max())
        assert frac < 0.05 and err <= 2.0 ** -7, f"{name}: {frac:.3f} of outputs differ, worst relative step {err:.2e}"


@pytest.mark.parametrize("case", FAST_CASES)
def test_conv3d_bf16_fast_path(dev, case):
    """bf16 MFMA fwd/dgrad vs (a) the generic fp32-matrix-core path on the GPU and (b) the CPU oracle.

    Weights are chosen bf16-representable so both GPU paths multiply identical operands and differ only in fp32
    summation order (then one bf16 rounding of the output)."""
    from video_vae_amd import ops
    from video_vae_amd._lib import lib
    ci, co, kt, kh, kw, (n, t, h, w) = case
    x = _bf16_exact((n, t, h, w, ci), 50, 1.0)
    k = _bf16_exact((kt, kh, kw, ci, co), 51, (kt * kh * kw * ci) ** -0.5)
    b = rnd((co,), 52, 0.1)
    gy = _bf16_exact((n, t, h, w, co), 53, 1.0)
    assert lib().vvae_conv3d_bf16_supported(ci, co, kt, kh, kw, ci, co, 0, 0) == 1
    assert lib().vvae_conv3d_bf16_supported(ci, co, kt, kh, kw, co, ci, 1, 0) == 1
    xg, kg, bg, gyg = x.to(dev, torch.bfloat16), k.to(dev), b.to(dev), gy.to(dev, torch.bfloat16)
    assert lib().vvae_conv3d_bf16_supported(ci, co, kt, kh, kw, ci, co, 2, 0) == 1
    y_fast = ops.conv3d_fwd_raw(xg, kg, bg)
    dx_fast = ops.conv3d_dgrad_raw(gyg, kg)
    dw_fast, db_fast = ops.conv3d_wgrad_raw(xg, gyg, tuple(k.shape))
    ops.force_generic_conv(True)
    try:
        y_gen = ops.conv3d_fwd_raw(xg, kg, bg)
        dx_gen = ops.conv3d_dgrad_raw(gyg, kg)
        dw_gen, db_gen = ops.conv3d_wgrad_raw(xg, gyg, tuple(k.shape))
    finally:
        ops.force_generic_conv(False)
    # wgrad: identical bf16 operands, fp32 accumulation in both paths -> agreement to fp32 summation-order noise
    assert_close_scaled(dw_fast, dw_gen, rel=2e-5, what="fast vs generic dw")
    assert_close_scaled(db_fast, db_gen, rel=2e-5, what="fast vs generic db")
    dw_again, _ = ops.conv3d_wgrad_raw(xg, gyg, tuple(k.shape))
    assert torch.equal(dw_again, dw_fast), "slab-reduced wgrad must be bitwise reproducible"
    assert_close(y_fast, y_gen, rtol=1e-2, atol=1e-2, what="fast vs generic y")
    assert_close(dx_fast, dx_gen, rtol=1e-2, atol=1e-2 * float(dx_gen.float().abs().max()), what="fast vs generic dx")
    frac = float((y_fast != y_gen).float().mean())
    assert frac < 0.05, f"{frac:.3f} of outputs differ between the two GPU paths (expected only rare 1-ulp flips)"
    xo = x.clone().requires_grad_(True)
    yo = O.conv3d_same(xo, k, b, torch.bfloat16)
    yo.backward(gy)
    assert_close(y_fast, yo, rtol=2e-2, atol=2e-2, what="fast vs oracle y")
    assert_close_scaled(dx_fast, xo.grad, rel=2e-2, what="fast vs oracle dx")


@pytest.mark.parametrize("shape", [(3, 7, 768), (2, 5, 4, 64), (4, 9, 96), (1, 3, 1024), (5, 40)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("use_bias", [True, False])
def test_layer_norm(dev, shape, dtype, use_bias):
    ops = _ops()
    c = shape[-1]
    x = (rnd(shape, 60) * 1.3 + 0.2).to(dtype).float()
    sc = 1 + 0.2 * rnd((c,), 61); bi = 0.1 * rnd((c,), 62) if use_bias else None
    gy = rnd(shape, 63).to(dtype).float()
    xo = x.clone().requires_grad_(True); so = sc.clone().requires_grad_(True)
    bo = bi.clone().requires_grad_(True) if use_bias else None
    yo = O.layer_norm(xo, so, bo, dtype)
    yo.backward(gy)
    xg = x.to(dev, dtype).requires_grad_(True); sg = sc.to(dev).requires_grad_(True)
    bg = bi.to(dev).requires_grad_(True) if use_bias else None
    assert ops.layer_norm_supported(xg)
    yg = ops.layer_norm(xg, sg, bg)
    yg.backward(gy.to(dev, dtype))
    r = 1e-3 if dtype == torch.float32 else 2e-2
    assert_close(yg, yo, rtol=r, atol=(1e-4 if dtype == torch.float32 else 2e-2), what="y")
    assert_close_scaled(xg.grad, xo.grad, rel=r, what="dx")
    assert_close_scaled(sg.grad, so.grad, rel=r, what="dscale")
    if use_bias:
        assert_close_scaled(bg.grad, bo.grad, rel=r, what="dbias")


@pytest.mark.parametrize("shape", [(3, 70, 768), (2, 9, 512), (130, 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layer_norm_fork_residual(dev, shape, dtype):
    """x + f(LN(x)) with the skip gradient added inside the LayerNorm-backward kernel == the unfused graph; and the
    skip-only / branch-only uses of the fork."""
    ops = _ops()
    c = shape[-1]
    x = (rnd(shape, 80) * 1.3 + 0.2).to(dev, dtype)
    sc = (1 + 0.2 * rnd((c,), 81)).to(dev); bi = (0.1 * rnd((c,), 82)).to(dev)
    w = (rnd(shape, 83)).to(dev, dtype); go = rnd(shape, 84).to(dev, dtype)
    f = lambda y: torch.tanh(y) * w

    def run(fused):
        xg = x.clone().requires_grad_(True); sg = sc.clone().requires_grad_(True); bg = bi.clone().requires_grad_(True)
        if fused:
            y, skip = ops.layer_norm_fork(xg, sg, bg)
            out = skip + f(y)
        else:
            out = xg + f(ops.layer_norm(xg, sg, bg))
        out.backward(go)
        return out, xg.grad, sg.grad, bg.grad
    o1, dx1, ds1, db1 = run(True)
    o2, dx2, ds2, db2 = run(False)
    assert torch.equal(o1, o2)
    assert torch.equal(ds1, ds2) and torch.equal(db1, db2)
    if dtype == torch.float32:
        assert_close(dx1, dx2, rtol=1e-6, atol=1e-6, what="dx fused vs unfused")
    else:
        assert torch.equal(dx1, dx2), "bf16: round(LN grad) + skip grad, rounded once more -- same as the separate add"
    xg = x.clone().requires_grad_(True)
    y, skip = ops.layer_norm_fork(xg, sc, bi)
    (skip * go).sum().backward()                                   # branch unused
    assert_close(xg.grad, go, what="skip-only gradient")
    xg = x.clone().requires_grad_(True)
    y, skip = ops.layer_norm_fork(xg, sc, bi)
    (y * go).sum().backward()                                      # skip unused
    xr = x.clone().requires_grad_(True)
    (ops.layer_norm(xr, sc, bi) * go).sum().backward()
    assert torch.equal(xg.grad, xr.grad)


@pytest.mark.parametrize("shape", [(1024, 2, 768), (8192, 128), (1000, 2, 64), (7, 12), (1, 4), (4096, 1536)])
def test_sum_rows(dev, shape):
    """Fold of per-workgroup partial rows: fixed order (bitwise reproducible) and equal to an fp64 column sum."""
    ops = _ops()
    part = rnd(shape, 90).to(dev)
    got = ops.sum_rows(part)
    assert got.shape == part.shape[1:]
    want = part.double().sum(0)
    assert_close(got.double(), want, rtol=1e-5, atol=1e-4, what="sum_rows")
    assert torch.equal(got, ops.sum_rows(part))


@pytest.mark.parametrize("shape", [(3, 70, 768), (2, 9, 512), (130, 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_add_layer_norm_fork(dev, shape, dtype):
    """(LN(skip + o), skip + o) in one kernel pass == add launch followed by the LayerNorm fork, forward and backward."""
    ops = _ops()
    c = shape[-1]
    skip = (rnd(shape, 85) * 1.3 + 0.2).to(dev, dtype); o = rnd(shape, 86).to(dev, dtype)
    sc = (1 + 0.2 * rnd((c,), 87)).to(dev); bi = (0.1 * rnd((c,), 88)).to(dev)
    w = rnd(shape, 89).to(dev, dtype); go = rnd(shape, 90).to(dev, dtype)

    def run(fused):
        sk = skip.clone().requires_grad_(True); oo = o.clone().requires_grad_(True)
        sg = sc.clone().requires_grad_(True); bg = bi.clone().requires_grad_(True)
        if fused:
            y, xs = ops.add_layer_norm_fork(sk, oo, sg, bg)
        else:
            y, xs = ops.layer_norm_fork(sk + oo, sg, bg)
        out = xs + torch.tanh(y) * w
        out.backward(go)
        return out, sk.grad, oo.grad, sg.grad, bg.grad
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]), "forward (sum rounded to the storage dtype, then normalised) is bit-identical"
    for i, nm in ((1, "dskip"), (2, "do")):
        if dtype == torch.float32:
            assert_close(a[i], b[i], rtol=1e-6, atol=1e-6, what=nm)
        else:
            assert torch.equal(a[i], b[i]), nm
    assert torch.equal(a[1], a[2])
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


def test_layer_norm_strided_head_view(dev):
    """q_norm on the q third of a fused QKV buffer, normalised in place of a gather copy (two-level row strides)."""
    ops = _ops()
    b, s, h, d = 2, 5, 4, 64
    qkv = rnd((b, s, 3 * h * d), 64)
    sc = 1 + 0.2 * rnd((d,), 65)
    q = qkv[..., :h * d].reshape(b, s, h, d)
    want = O.layer_norm(q, sc, None)
    qg = qkv.to(dev)[..., h * d:2 * h * d].unflatten(-1, (h, d))           # k third: non-contiguous view
    want = O.layer_norm(qkv[..., h * d:2 * h * d].reshape(b, s, h, d), sc, None)
    assert not qg.is_contiguous()
    got = ops.layer_norm(qg, sc.to(dev), None)
    assert_close(got, want, what="strided LN")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_temporal_attention_strided_layout(dev, dtype):
    """inner = hw: (b, t, hw, C) in, same out, equal to the transposed contiguous form, forward and backward."""
    from einops import rearrange
    ops = _ops()
    b, t, hw, heads, d = 2, 16, 6, 4, 64
    qkv = rnd((b, t, hw, 3 * heads * d), 70).to(dev, dtype)
    qs = (1 + 0.2 * rnd((d,), 71)).to(dev); ks = (1 + 0.2 * rnd((d,), 72)).to(dev)
    go = rnd((b, t, hw, heads * d), 73).to(dev, dtype)
    cos, sin = OL.rope_tables(d, 64)
    cos, sin = cos.to(dev), sin.to(dev)
    m8 = (torch.arange(t)[None, :] < torch.tensor([t, 9])[:, None]).to(torch.uint8).to(dev)       # (b, t)
    x1 = qkv.clone().requires_grad_(True); q1 = qs.clone().requires_grad_(True); k1 = ks.clone().requires_grad_(True)
    o1 = ops.temporal_attention_core(x1, q1, k1, cos, sin, m8, hw, heads, 1e-6, inner=hw)
    o1.backward(go)
    x2 = qkv.clone().requires_grad_(True); q2 = qs.clone().requires_grad_(True); k2 = ks.clone().requires_grad_(True)
    o2 = ops.temporal_attention_core(rearrange(x2, "b t hw c -> (b hw) t c").contiguous(), q2, k2, cos, sin, m8, hw, heads)
    o2 = rearrange(o2, "(b hw) t c -> b t hw c", b=b)
    o2.backward(go)
    assert torch.equal(o1, o2)
    assert torch.equal(x1.grad, x2.grad)
    assert_close_scaled(q1.grad, q2.grad, rel=1e-5)
    assert_close_scaled(k1.grad, k2.grad, rel=1e-5)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 1000), (768, 1536, 4096), (512, 128, 33), (256, 256, 32), (512, 768, 2080),
                                   (768, 512, 16384), (256, 512, 96), (768, 96, 16384), (96, 768, 16384), (96, 96, 500), (8, 200, 77), (136, 24, 64)])
@pytest.mark.parametrize("big", [True, False])
def test_gemm_tn_weight_gradient(dev, m, n, k, big):
    """dW = X^T dY and db = colsum(dY) from the split-K HIP kernels vs fp32 torch on the same bf16 operands.
    big=True: 256 x 256 tiles (LDS-DMA ring, staggered wave halves) where M, N, K allow; False: the 128 x 128 kernel.  M, N that are
    multiples of 8 but not of 128 (the 96-wide latent heads) run on edge tiles of the 128 x 128 kernel: zero columns in, guarded stores out."""
    from video_vae_amd._lib import lib
    ops = _ops()
    a = rnd((k, m), 80).to(dev, torch.bfloat16)
    b = rnd((k, n), 81).to(dev, torch.bfloat16)
    assert ops.gemm_tn_supported(a, b)
    lib().vvae_gemm_tn_use_big_tiles(1 if big else 0)
    try:
        c, db = ops.gemm_tn(a, b)
        c2, _ = ops.gemm_tn(a, b)
        c3, none = ops.gemm_tn(a, b, False)
    finally:
        lib().vvae_gemm_tn_use_big_tiles(1)
    want = a.float().t() @ b.float()
    assert_close_scaled(c, want, rel=1e-4, what="dW")
    assert_close_scaled(db, b.float().sum(0), rel=1e-4, what="db")
    assert torch.equal(c, c2), "slab reduction must be bitwise reproducible"
    assert none is None and torch.equal(c, c3)


def test_gemm_tn_grouped_deferred(dev):
    """Parked Linear weight gradients multiplied in one grouped launch (whole-K tiles, no slabs) straight into the flat gradient
    slots == the per-product split-K kernel; bias sums included; the optimizer is told which slots are already in place."""
    import types
    ops = _ops()
    k = 1024
    shapes = [(768, 1536), (512, 768), (1536, 768), (256, 256), (768, 768), (1536, 1536), (512, 512), (768, 512), (256, 1536), (1536, 256),
              (1536, 1536)]
    items, want = [], []
    marked = []
    opt = types.SimpleNamespace(mark_external=lambda p: marked.append(p))
    for i, (m, n) in enumerate(shapes):
        a = rnd((k, m), 120 + i).to(dev, torch.bfloat16); b = rnd((k, n), 140 + i).to(dev, torch.bfloat16)
        kern = torch.nn.Parameter(torch.zeros(m, n, device=dev)); bias = torch.nn.Parameter(torch.zeros(n, device=dev))
        kern.gview = torch.full((m, n), 7.0, device=dev); bias.gview = torch.full((n,), 7.0, device=dev)
        items.append((a, b, kern, bias))
        want.append(ops.gemm_tn(a, b, True))
    assert sum((m // 256) * (n // 256) for m, n in shapes) >= ops.GROUP_MIN_TILES
    ops.WGRAD_QUEUE[0] = []
    try:
        for a, b, kern, bias in items:
            assert ops.wgrad_deferrable(a, b, kern, bias)
    finally:
        ops.WGRAD_QUEUE[0] = None
    q = ops._WgradQueue(opt)
    for it in items:
        q.append(it)
    q.flush()
    for (a, b, kern, bias), (dw, db) in zip(items, want):
        assert_close_scaled(kern.gview, dw, rel=2e-5, what="grouped dW vs split-K dW")
        assert_close_scaled(bias.gview, db, rel=2e-5, what="grouped db")
        assert_close_scaled(kern.gview, a.float().t() @ b.float(), rel=1e-4, what="grouped dW vs fp32")
    assert len(marked) == 2 * len(items)
    # a short queue takes the per-product path and lands in the same slots
    for a, b, kern, bias in items[:2]:
        kern.gview.fill_(3.0); bias.gview.fill_(3.0)
    ops.flush_wgrad(items[:2], opt)
    for (a, b, kern, bias), (dw, db) in zip(items[:2], want[:2]):
        assert torch.equal(kern.gview, dw) and torch.equal(bias.gview, db)


@pytest.mark.parametrize("ci,co,shape", [(32, 16, (2, 3, 6, 10)), (64, 32, (1, 2, 5, 7)), (128, 64, (1, 2, 4, 8)), (32, 16, (1, 5, 32, 40)),
                                         (128, 64, (2, 4, 16, 16))])
def test_conv_transpose_wgrad_bf16(dev, ci, co, shape):
    """bf16 matrix-core ConvTranspose weight + bias gradient (one wave per tap, slab reduction) vs the generic fp32-matrix-core
    path + column sum, and bitwise reproducible."""
    from video_vae_amd import ops
    n, t, h, w = shape
    x = rnd((n, t, h, w, ci), 110).to(dev, torch.bfloat16)
    gy = rnd((n, t, 2 * h, 2 * w, co), 111).to(dev, torch.bfloat16)
    ks = (1, 2, 2, ci, co)
    dw, db = ops.convt_wgrad_db_raw(x, gy, ks)
    dw2, db2 = ops.convt_wgrad_db_raw(x, gy, ks)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    ops.force_generic_conv(True)
    try:
        dw_gen, db_gen = ops.convt_wgrad_db_raw(x, gy, ks)
    finally:
        ops.force_generic_conv(False)
    assert_close_scaled(dw, dw_gen, rel=2e-5, what="dw fast vs generic")
    assert_close_scaled(db, db_gen, rel=2e-5, what="db fast vs generic")


@pytest.mark.parametrize("cin", [16, 12])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 9, 11), (1, 1, 1, 1), (1, 4, 64, 80)])
def test_conv_pointwise_stream_kernels(dev, cin, dtype, shape):
    """1x1x1 convs onto 3 channels (final_conv, un-embedding down-projection): HBM-stream kernels vs the generic matrix-core
    path and vs the oracle, forward, input gradient, weight and bias gradients (fixed-order partial folds: reproducible)."""
    from video_vae_amd import ops
    n, t, h, w = shape
    x = rnd((n, t, h, w, cin), 100).to(dtype).float()
    k = rnd((1, 1, 1, cin, 3), 101, cin ** -0.5)
    b = rnd((3,), 102, 0.1)
    gy = rnd((n, t, h, w, 3), 103).to(dtype).float()
    xg, kg, bg, gyg = x.to(dev, dtype), k.to(dev), b.to(dev), gy.to(dev, dtype)
    y = ops.conv3d_fwd_raw(xg, kg, bg)
    dx = ops.conv3d_dgrad_raw(gyg, kg)
    dw, db = ops.conv3d_wgrad_raw(xg, gyg, tuple(k.shape))
    dw2, _ = ops.conv3d_wgrad_raw(xg, gyg, tuple(k.shape))
    assert torch.equal(dw, dw2)
    ops.force_generic_conv(True)
    try:
        y_gen = ops.conv3d_fwd_raw(xg, kg, bg)
        dx_gen = ops.conv3d_dgrad_raw(gyg, kg)
        dw_gen, db_gen = ops.conv3d_wgrad_raw(xg, gyg, tuple(k.shape))
    finally:
        ops.force_generic_conv(False)
    xo = x.clone().requires_grad_(True); ko = k.clone().requires_grad_(True); bo = b.clone().requires_grad_(True)
    yo = O.conv3d_same(xo, ko, bo, dtype)
    yo.backward(gy)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert_close(y, yo, rtol=tol, atol=tol, what="y vs oracle")
    assert_close(y, y_gen, rtol=tol, atol=tol, what="y vs generic")
    assert_close_scaled(dx, xo.grad, rel=tol, what="dx vs oracle")
    assert_close_scaled(dx, dx_gen, rel=tol, what="dx vs generic")
    tolw = 2e-4 if dtype == torch.float32 else 1e-2          # the oracle's bf16 emulation rounds the parameter gradient
    assert_close_scaled(dw, ko.grad, rel=tolw, floor=1e-6, what="dw vs oracle")
    assert_close_scaled(dw, dw_gen, rel=2e-4, floor=1e-6, what="dw vs generic")
    assert_close_scaled(db, bo.grad, rel=tolw, floor=1e-6, what="db vs oracle")
    assert_close_scaled(db, db_gen, rel=2e-4, floor=1e-6, what="db vs generic")


@pytest.mark.parametrize("m,n,k", [(256, 192, 64), (512, 768, 512), (256, 128, 128), (1024, 1536, 768), (256, 512, 192),
                                   # more than 256 tiles: persistent workgroups walk 2 (production shape), 3 and 2 (128-wide) tiles,
                                   # the shortest even K (next tile requested in k-step 1 of 2), and an odd K (one tile per workgroup)
                                   (16384, 1536, 768), (16384, 2304, 256), (16384, 1024, 256), (8192, 3072, 128), (16384, 1536, 192)])
def test_gemm_nt_linear_forms(dev, m, n, k):
    """C = epi(A B^T + bias) (LDS-DMA ring NT GEMM) vs fp32 torch: plain, + residual, SiLU (+ saved pre-activation) and
    * silu'(h); the fused tails act on the bf16-rounded linear output, i.e. equal Linear followed by the separate op."""
    import torch.nn.functional as F
    ops = _ops()
    a = rnd((m, k), 95).to(dev, torch.bfloat16)
    b = (rnd((n, k), 96) / k ** 0.5).to(dev, torch.bfloat16)
    bias = rnd((n,), 97).to(dev)
    res = rnd((m, n), 98).to(dev, torch.bfloat16)
    assert ops.gemm_nt_supported(a, b)
    ref = a.float() @ b.float().t() + bias
    c = ops.gemm_nt(a, b, bias)
    assert_close(c, ref, rtol=1e-2, atol=1e-2, what="linear")
    assert float((c != ref.to(torch.bfloat16)).float().mean()) < 5e-3, "only accumulation-order flips of the last bf16 bit"
    c1 = ops.gemm_nt(a, b, bias, res, ops.EPI_RES)
    assert torch.equal(c1, (c.float() + res.float()).to(torch.bfloat16)), "residual add on the rounded output"
    c2, h = ops.gemm_nt(a, b, bias, None, ops.EPI_SILU)
    assert torch.equal(h, c)
    assert_close(c2, F.silu(h.float()), rtol=1e-2, atol=1e-2, what="silu")
    base = ops.gemm_nt(a, b, None)
    c3 = ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU)
    sg = torch.sigmoid(res.float())
    assert_close(c3, base.float() * (sg * (1 + res.float() * (1 - sg))), rtol=1e-2, atol=1e-2, what="dsilu")
    assert torch.equal(c, ops.gemm_nt(a, b, bias)), "deterministic"
    if m * n > 256 * 256 * 192:                                    # persistent launch form == one tile per workgroup, bit for bit
        from video_vae_amd._lib import lib
        assert lib().vvae_gemm_nt_persistent(0) == 0
        try:
            assert torch.equal(c, ops.gemm_nt(a, b, bias)) and torch.equal(c3, ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU))
            c2n, hn = ops.gemm_nt(a, b, bias, None, ops.EPI_SILU)
            assert torch.equal(c2, c2n) and torch.equal(h, hn) and torch.equal(c1, ops.gemm_nt(a, b, bias, res, ops.EPI_RES))
        finally:
            lib().vvae_gemm_nt_persistent(1)
        try:                                                       # the L2 prefetch of the token panel changes timing only
            for dist in (0, 2, 5):
                assert lib().vvae_gemm_nt_prefetch(dist) == 0
                assert torch.equal(c, ops.gemm_nt(a, b, bias)) and torch.equal(c3, ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU))
                c2n, hn = ops.gemm_nt(a, b, bias, None, ops.EPI_SILU)
                assert torch.equal(c2, c2n) and torch.equal(h, hn)
        finally:
            lib().vvae_gemm_nt_prefetch(3)


@pytest.mark.parametrize("m,n,k", [(16384, 1536, 768), (16384, 768, 1536), (16384, 512, 768), (16384, 768, 512), (512, 384, 128), (256, 1536, 192),
                                     (1024, 2048, 256), (768, 192, 832), (16384, 1536, 192)])
def test_gemm_pp_matches_gemm_nt_bitwise_and_fp32(dev, m, n, k):
    """The second NT GEMM form (csrc/gemm_pp.hip: staging from the read phases, three-slot token ring, per-wave epilogue) against the first
    (gemm_nt.hip) BIT FOR BIT on all four epilogues, with and without bias, and against the fp32 product: the trunk's shapes (two tiles per
    workgroup in one k-tile stream at N = 1536; one tile at 768 / 512), the smallest K (two k-tiles), K an odd number of k-tiles (the three-slot
    ring wraps out of step with the two-slot one), a tile count that is not a multiple of 256 (one tile per workgroup), N = 2048 (the whole
    LDS bias buffer of the 128-column form), pitched A / residual / output views, and twice (deterministic)."""
    ops = _ops()
    a_full = rnd((m, k + 64), 195).to(dev, torch.bfloat16)
    a = a_full[:, :k]                                              # row pitch k + 64
    b = (rnd((n, k), 196) / k ** 0.5).to(dev, torch.bfloat16)
    bias = rnd((n,), 197).to(dev)
    res = rnd((m, n), 198).to(dev, torch.bfloat16)
    assert ops.gemm_nt_supported(a, b) and ops.lib().vvae_gemm_pp_supported(m, n, k, a.stride(0), b.stride(0), n) == 1
    ref = a.float() @ b.float().t() + bias
    for epi, kw in ((ops.EPI_NONE, dict(bias=bias)), (ops.EPI_NONE, dict()), (ops.EPI_RES, dict(bias=bias, res=res)), (ops.EPI_SILU, dict(bias=bias)),
                    (ops.EPI_MUL_DSILU, dict(res=res))):
        want = ops.gemm_nt(a, b, epi=epi, form="nt", **kw)
        got = ops.gemm_nt(a, b, epi=epi, form="pp", **kw)
        again = ops.gemm_nt(a, b, epi=epi, form="pp", **kw)
        for w, g_, g2 in zip(*[x if isinstance(x, tuple) else (x,) for x in (want, got, again)]):
            assert torch.equal(w, g_), (epi, sorted(kw), int((w != g_).sum()))
            assert torch.equal(g_, g2)
    c = ops.gemm_nt(a, b, bias, form="pp")
    assert_close(c, ref, rtol=1e-2, atol=1e-2, what="linear")
    assert float((c != ref.to(torch.bfloat16)).float().mean()) < 5e-3, "only accumulation-order flips of the last bf16 bit"


def test_gemm_pp_declines_what_it_cannot_take(dev):
    ops = _ops()
    L = ops.lib()
    assert L.vvae_gemm_pp_supported(16384, 1536, 64, 64, 64, 1536) == 0          # one k-tile: the stream needs two
    assert L.vvae_gemm_pp_supported(16384, 1728, 768, 768, 768, 1728) == 0        # N = 9 x 192 > 1536: the bias vector would not fit the LDS left over
    assert L.vvae_gemm_pp_supported(16384, 2176, 768, 768, 768, 2176) == 0        # N = 17 x 128 > 2048
    assert L.vvae_gemm_pp_supported(16380, 768, 768, 768, 768, 768) == 0
    a = torch.zeros((256, 128), device=dev, dtype=torch.bfloat16)
    with pytest.raises(ops.VvaeError):
        ops.gemm_nt(a, torch.zeros((100, 128), device=dev, dtype=torch.bfloat16), form="pp")


@pytest.mark.parametrize("ci,co,shape", [(32, 16, (2, 3, 6, 10)), (64, 32, (1, 2, 5, 7)), (128, 64, (1, 2, 4, 8))])
def test_conv_transpose_bf16_fast_path(dev, ci, co, shape):
    """bf16 MFMA ConvTranspose fwd/dgrad (weights in registers) vs the generic fp32-matrix-core path and the oracle."""
    from video_vae_amd import ops
    from video_vae_amd._lib import lib
    n, t, h, w = shape
    x = _bf16_exact((n, t, h, w, ci), 90, 1.0)
    k = _bf16_exact((1, 2, 2, ci, co), 91, (4 * ci) ** -0.5)
    b = rnd((co,), 92, 0.1)
    gy = _bf16_exact((n, t, 2 * h, 2 * w, co), 93, 1.0)
    assert lib().vvae_convt_bf16_supported(ci, co, ci, co) == 1
    xg, kg, bg, gyg = x.to(dev, torch.bfloat16), k.to(dev), b.to(dev), gy.to(dev, torch.bfloat16)
    y_fast = ops.convt_fwd_raw(xg, kg, bg)
    dx_fast = ops.convt_dgrad_raw(gyg, kg)
    ops.force_generic_conv(True)
    try:
        y_gen = ops.convt_fwd_raw(xg, kg, bg)
        dx_gen = ops.convt_dgrad_raw(gyg, kg)
    finally:
        ops.force_generic_conv(False)
    assert_close(y_fast, y_gen, rtol=1e-2, atol=1e-2, what="fast vs generic y")
    assert_close(dx_fast, dx_gen, rtol=1e-2, atol=1e-2 * float(dx_gen.float().abs().max()), what="fast vs generic dx")
    xo = x.clone().requires_grad_(True)
    yo = O.conv_transpose_1x2x2(xo, k, b, torch.bfloat16)
    yo.backward(gy)
    assert_close(y_fast, yo, rtol=2e-2, atol=2e-2, what="fast vs oracle y")
    assert_close_scaled(dx_fast, xo.grad, rel=2e-2, what="fast vs oracle dx")
    # channel-slice destination (concat elision): write the up-sampled half of a 2*co buffer in place
    buf = torch.zeros((n, t, 2 * h, 2 * w, 2 * co), dtype=torch.bfloat16, device=dev)
    ops.convt_fwd_raw(xg, kg, bg, out=buf[..., :co])
    assert torch.equal(buf[..., :co], y_fast) and float(buf[..., co:].abs().max()) == 0


@pytest.mark.parametrize("rows,mlp,out", [(1024, 1536, 768), (512, 1536, 768), (96, 64, 32)])
def test_silu_linear_fused_backward(dev, rows, mlp, out):
    """linear2(silu(h)) as one node (dh = (dy @ W2^T) * silu'(h) in a GEMM epilogue) vs the three-node chain; reference
    train/layers.py:186-189."""
    import video_vae_amd as V
    from video_vae_amd import layers as LY, optim

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear2 = LY.Linear(mlp, out, V.Rngs(1))
    m = Two().to(dev)
    opt = optim.Optimizer(m, 1e-3)                                 # gives the parameters their bf16 shadows
    with torch.no_grad():
        m.linear2.bias.copy_(rnd((out,), 3, 0.1).to(dev))
    opt.refresh_shadow()
    h = rnd((rows, mlp), 70, 1.5).to(dev, torch.bfloat16)
    gy = rnd((rows, out), 71, 1.0).to(dev, torch.bfloat16)
    res = []
    for fused in (True, False):
        hh = h.clone().requires_grad_(True)
        opt.zero_grad()
        y = LY.silu_linear(hh, m.linear2) if fused else m.linear2(torch.nn.functional.silu(hh))
        y.backward(gy)
        for b in range(len(opt.buckets)):                          # gradients live in the optimizer's flat buffer
            if not opt.landed[b]:
                opt._land(b)
        res.append((y.detach(), hh.grad, m.linear2.kernel.gview.clone(), m.linear2.bias.gview.clone()))
    (y1, dh1, dw1, db1), (y0, dh0, dw0, db0) = res
    # the node's activation is the library's own SiLU stream kernel (v_rcp_f32 sigmoid): a rare 1-ulp difference from the framework's
    assert_close(y1, y0, rtol=1e-2, atol=1e-2, what="y fused vs chain")
    assert float((y1 != y0).float().mean()) < 0.02
    assert_close_scaled(dh1, dh0, rel=1e-2, what="dh fused vs chain")        # chain rounds dy @ W2^T to bf16 before the multiply
    assert_close_scaled(dw1, dw0, rel=1e-3, what="dW2")
    assert_close_scaled(db1, db0, rel=2e-5, what="db2")
    # against fp32 math on the same bf16 operands
    hf = h.float().requires_grad_(True)
    yf = torch.nn.functional.silu(hf).to(torch.bfloat16).float() @ m.linear2.kernel.bf16.float() + m.linear2.bias.bf16.float()
    yf.backward(gy.float())
    assert_close_scaled(dh1, hf.grad, rel=1e-2, what="dh fused vs fp32")


def test_copy_grouped(dev):
    """Grouped landing copy: ragged sizes, unaligned ranges, more than one launch."""
    from video_vae_amd import ops
    sizes = [1, 3, 4, 1023, 1024, 1025, 4099, 70001] + [17 + 5 * i for i in range(70)]
    flat_s = rnd((sum(sizes) + 8,), 90).to(dev)
    flat_d = torch.zeros(sum(sizes) + 8, device=dev)
    srcs, dsts, o = [], [], 3                                     # offset 3: ranges that are not 16-byte aligned
    for n in sizes:
        srcs.append(flat_s[o:o + n]); dsts.append(flat_d[o:o + n]); o += n
    assert all(ops.copy_grouped_ok(d, s) for d, s in zip(dsts, srcs))
    ops.copy_grouped(dsts, srcs)
    assert torch.equal(flat_d[3:o], flat_s[3:o])
    assert float(flat_d[:3].abs().max()) == 0 and float(flat_d[o:].abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upblock_concat_elision_matches_cat(dev, dtype):
    """UpBlock3D with the joint buffer (up-conv and the encoder's GroupNorm+SiLU write the two channel halves in place,
    ops.join_channels) == the reference's concat form (train/unet.py:76-83): outputs and all gradients."""
    import video_vae_amd as V
    from video_vae_amd import ops, unet as U
    c = 16
    up = U.UpBlock3D(2 * c, c, V.Rngs(4), dtype=dtype).to(dev)
    blk = U.ConvBlock3D(c, c, 3, V.Rngs(5), dtype=dtype).to(dev)            # stands for the encoder's conv2 (skip producer)
    x = rnd((1, 3, 6, 10, 2 * c), 80).to(dev, dtype)
    s_in = rnd((1, 3, 12, 20, c), 81).to(dev, dtype)
    gy = rnd((1, 3, 12, 20, c), 82).to(dev, dtype)
    res = []
    for fused in (True, False):
        xx, ss = x.clone().requires_grad_(True), s_in.clone().requires_grad_(True)
        up.zero_grad(); blk.zero_grad()
        if fused:
            joint = torch.empty((1, 3, 12, 20, 2 * c), dtype=dtype, device=dev)
            skip = blk(ss, out=joint[..., c:])
            y = up(xx, skip, joint)
        else:
            y = up(xx, blk(ss), None)
        (y.float() * gy.float()).sum().backward()
        res.append([y.detach(), xx.grad, ss.grad] + [p.grad.clone() for p in list(up.parameters()) + list(blk.parameters())])
    assert torch.equal(res[0][0], res[1][0])
    for k, (a, b) in enumerate(zip(*res)):                  # gradients: the generic fp32 weight-gradient path sums with float atomics
        assert_close_scaled(a, b, rel=2e-5 if dtype == torch.float32 else 1e-6, what=f"tensor {k}")


@pytest.mark.parametrize("ci,co,kh,hw", [(16, 16, 3, 256), (32, 16, 3, 256), (16, 16, 7, 256), (32, 32, 3, 128), (64, 32, 3, 128)])
def test_conv3d_full_size_properties(dev, ci, co, kh, hw):
    """BASELINE's full extent (B=4, 16 frames, 256^2 / 128^2): properties that need no oracle.  Scaling an operand by 2 is exact
    in bf16 and in every fp32 partial sum, so fwd / dgrad / wgrad must scale bitwise; the rolling and per-frame kernels must agree
    bitwise; a clip shifted by one frame gives the shifted output away from the temporal border."""
    from video_vae_amd import ops
    from video_vae_amd._lib import lib
    g = torch.Generator(device="cpu").manual_seed(ci * 131 + co * 7 + kh)
    x = (torch.randn((4, 16, hw, hw, ci), generator=g) * 0.5).to(dev, torch.bfloat16)
    gy = (torch.randn((4, 16, hw, hw, co), generator=g) * 0.5).to(dev, torch.bfloat16)
    k = (torch.randn((3, kh, kh, ci, co), generator=g) * (3 * kh * kh * ci) ** -0.5).to(dev)
    zero_b = torch.zeros(co, device=dev)
    y = ops.conv3d_fwd_raw(x, k, zero_b)
    dx = ops.conv3d_dgrad_raw(gy, k)
    dw, db = ops.conv3d_wgrad_raw(x, gy, tuple(k.shape))
    assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all() and torch.isfinite(dw).all()
    assert torch.equal(ops.conv3d_fwd_raw(x * 2, k, zero_b), y * 2)
    assert torch.equal(ops.conv3d_dgrad_raw(gy * 2, k), dx * 2)
    dw2, db2 = ops.conv3d_wgrad_raw(x, gy * 2, tuple(k.shape))
    assert torch.equal(dw2, dw * 2) and torch.equal(db2, db * 2)
    try:
        lib().vvae_conv3d_roll_config(0, 0)
        assert torch.equal(ops.conv3d_fwd_raw(x, k, zero_b), y)
        assert torch.equal(ops.conv3d_dgrad_raw(gy, k), dx)
    finally:
        lib().vvae_conv3d_roll_config(1, 0)
    xs = torch.zeros_like(x)
    xs[:, 1:] = x[:, :-1]
    ys = ops.conv3d_fwd_raw(xs, k, zero_b)
    assert torch.equal(ys[:, 1:-1], y[:, :-2])              # frame 15 of the shifted clip misses x[15]: excluded
    # db = column sums of gy: compare with a float64 reduction
    assert_close_scaled(db, gy.double().sum((0, 1, 2, 3)).float(), rel=1e-5, what="db")


def test_unpatch_pad_matches_rearrange_then_pad(dev):
    """PatchUnEmbedding.forward_padded (one strided copy: un-patchify + zero channel pad; the 1x1x1 down-projection through
    zero-padded weight rows) == forward() followed by F.pad (reference train/layers.py:44-55): outputs and every gradient."""
    import torch.nn.functional as F
    import video_vae_amd as V
    from video_vae_amd import layers as LY, optim
    pu = LY.PatchUnEmbedding(32, 48, 3, 8, 4, V.Rngs(7)).to(dev)          # 12 feature channels -> padded to 16
    optim.Optimizer(pu, 1e-3)                                            # bf16 shadows for the Linear layers
    x = rnd((2, 3, 24, 192), 95).to(dev, torch.bfloat16)
    g_feat = rnd((2, 3, 32, 48, 16), 96).to(dev, torch.bfloat16)
    g_coarse = rnd((2, 3, 32, 48, 3), 97).to(dev, torch.bfloat16)
    res = []
    for fused in (True, False):
        xx = x.clone().requires_grad_(True)
        pu.zero_grad()
        if fused:
            feat, coarse = pu.forward_padded(xx)
        else:
            feat, coarse = pu(xx)
            feat = F.pad(feat, (0, 4))
        assert feat.shape[-1] == 16 and float(feat[..., 12:].float().abs().max()) == 0
        (feat.float() * g_feat.float()).sum().backward(retain_graph=True)
        (coarse.float() * g_coarse.float()).sum().backward()
        res.append([feat.detach(), coarse.detach(), xx.grad.clone()])
    (f1, c1, gx1), (f0, c0, gx0) = res
    assert torch.equal(f1, f0)
    assert_close(c1, c0, rtol=1e-2, atol=1e-2, what="coarse")
    assert_close_scaled(gx1, gx0, rel=2e-2, what="dx")


@pytest.mark.parametrize("ci,co,shape", [(16, 16, (2, 5, 20, 24)), (16, 32, (1, 4, 9, 33)), (32, 16, (2, 3, 17, 16)), (32, 32, (1, 6, 18, 30)),
                                         (64, 32, (1, 2, 8, 16)), (32, 64, (2, 3, 12, 20)), (64, 64, (1, 5, 9, 17)), (64, 128, (1, 2, 8, 16)),
                                         (128, 128, (2, 3, 6, 16)), (128, 64, (1, 4, 10, 18)), (48, 16, (1, 2, 8, 16))])
def test_conv_block_gn_statistics_from_conv_epilogue(dev, ci, co, shape):
    """ConvBlock3D where the conv kernel (rolling; round 3: also the deep rolling kernel of the 64 / 128-channel layers, whose workgroups cover
    whole GroupNorm groups of their channel block) sums its own rounded outputs per GroupNorm group (no statistics pass over the tensor)
    == the two-kernel form (vvae_gn_stats); (48, 16) is not eligible and must take the two-kernel form itself.
    Reference train/unet.py:13-30."""
    import video_vae_amd as V
    from video_vae_amd import ops, unet as U
    from video_vae_amd._lib import lib
    blk = U.ConvBlock3D(ci, co, 3, V.Rngs(9)).to(dev)
    with torch.no_grad():
        blk.conv.bias.copy_(rnd((co,), 3, 0.2).to(dev)); blk.norm.scale.copy_((1 + rnd((co,), 4, 0.2)).to(dev)); blk.norm.bias.copy_(rnd((co,), 5, 0.2).to(dev))
    n, t, h, w = shape
    x = rnd((n, t, h, w, ci), 100).to(dev, torch.bfloat16)
    gy = rnd((n, t, h, w, co), 101).to(dev, torch.bfloat16)
    eligible = ops.conv3d_gn_blocks(x, blk.conv.kernel, blk.norm.num_groups)
    assert (eligible > 0) == (ci != 48)
    # the fused partial sums against a direct fp64 reduction of the conv output
    if eligible:
        yc, part = ops.conv3d_fwd_gn_raw(x, blk.conv.kernel.detach(), blk.conv.bias.detach(), blk.norm.num_groups, eligible)
        cpg = co // blk.norm.num_groups
        ref = yc.double().reshape(n, -1, blk.norm.num_groups, cpg)
        assert_close_scaled(part.double().sum(1)[..., 0], ref.sum((1, 3)), rel=1e-5, what="group sums")
        assert_close_scaled(part.double().sum(1)[..., 1], (ref * ref).sum((1, 3)), rel=1e-5, what="group sums of squares")
    res = []
    for fused in (True, False):
        try:
            lib().vvae_conv3d_roll_config(1 if fused else 0, 0)
            lib().vvae_conv3d_deep_config(1 if fused else 0)
            xx = x.clone().requires_grad_(True)
            blk.zero_grad()
            y = blk(xx)
            y.backward(gy)
            res.append([y.detach(), xx.grad] + [p.grad.clone() for p in blk.parameters()])
        finally:
            lib().vvae_conv3d_roll_config(1, 0)
            lib().vvae_conv3d_deep_config(1)
    for k, (a, b) in enumerate(zip(*res)):
        assert_close_scaled(a, b, rel=2e-2 if k < 2 else 1e-3, what=f"tensor {k}")
    assert float((res[0][0] != res[1][0]).float().mean()) < 0.02          # the statistics differ in fp32 summation order only


@pytest.mark.parametrize("o", [1, 2, 4, 8, 16, 32])
def test_xor_lane_exchange_selftest(dev, o):
    """The VALU-only lane exchange (DPP / v_permlane16_swap / v_permlane32_swap) under every wave-level reduction of the library:
    y[i] == x[i ^ o] inside each group of 64 lanes, bit for bit, for fp32 and for fp64 values."""
    import ctypes
    from video_vae_amd._lib import lib, check
    n = 64 * 7
    x = torch.arange(n, dtype=torch.float32, device=dev) * 1.25 + 3.0
    y = torch.empty_like(x)
    yd = torch.empty(n, dtype=torch.float64, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    check(lib().vvae_selftest_xor_lane(p(x), p(y), p(yd), n, o, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "selftest")
    idx = torch.arange(n, device=dev) ^ o
    assert torch.equal(y, x[idx])
    assert torch.equal(yd, x[idx].double() * 1.000000001)


@pytest.mark.parametrize("a,heads,inner,masked,t", [(6, 8, 1, True, 16), (5, 3, 1, False, 16), (8, 8, 4, True, 16), (12, 2, 6, False, 16),
                                                    (6, 8, 1, True, 32), (5, 3, 1, False, 32), (8, 4, 4, True, 32), (12, 2, 6, False, 32),
                                                    (4, 2, 2, True, 64), (3, 1, 1, False, 64), (6, 3, 1, True, 64)])
def test_temporal_attention_matrix_core_kernels(dev, a, heads, inner, masked, t):
    """bf16, head_dim 64, T = 16 (the production temporal shape, attn_temporal_mfma.hip) and T = 32 / 64 (the later curriculum stages
    and config C5, attn_temporal_mfma32.hip) run on the matrix-core kernels: against the oracle, and against the VALU kernels they
    replace (test hooks), contiguous and strided (FactoredAttention's (b, t, hw, c)) sequence layouts, item counts that do not fill
    the last workgroup, shared mask rows."""
    ops = _ops()
    from video_vae_amd._lib import lib
    d, dtype = 64, torch.bfloat16
    hook = lib().vvae_temporal_attn_mfma_enable if t == 16 else lib().vvae_temporal_attn_mfma32_enable
    bsz = a // inner
    shape = (a, t, 3 * heads * d) if inner == 1 else (bsz, t, inner, 3 * heads * d)
    qkv = rnd(shape, 19).to(dtype).float()
    qs = 1 + 0.2 * rnd((d,), 20); ks = 1 + 0.2 * rnd((d,), 21)
    go = rnd(shape[:-1] + (heads * d,), 22).to(dtype).float()
    nm = a // 2 if inner > 1 else a                                   # strided case: two sequences share a mask row (mask_div = 2)
    mask = None
    if masked:
        lens = torch.tensor([max(1, t - (i * 5) % t) for i in range(nm)])
        mask = (torch.arange(t)[None, :] < lens[:, None])
    cos, sin = OL.rope_tables(d, 64)
    # oracle on (a, t, c) sequences
    seq = (lambda z: z) if inner == 1 else (lambda z: z.permute(0, 2, 1, 3).reshape(a, t, -1))
    unseq = (lambda z: z) if inner == 1 else (lambda z: z.reshape(bsz, inner, t, -1).permute(0, 2, 1, 3))
    xo = qkv.clone().requires_grad_(True); qso = qs.clone().requires_grad_(True); kso = ks.clone().requires_grad_(True)
    mo = None
    if masked:
        mo = mask.repeat_interleave(a // nm, dim=0).reshape(a, 1, 1, t)
    yo = unseq(_attn_ref(seq(xo), qso, kso, mo, heads, 64, dtype))
    yo.backward(go)
    m8 = mask.to(torch.uint8).to(dev) if masked else None

    def run(mfma):
        hook(1 if mfma else 0)
        try:
            xg = qkv.to(dev, dtype).requires_grad_(True); qsg = qs.to(dev).requires_grad_(True); ksg = ks.to(dev).requires_grad_(True)
            yg = ops.temporal_attention_core(xg, qsg, ksg, cos.to(dev), sin.to(dev), m8, a // nm, heads, inner=inner)
            yg.backward(go.to(dev, dtype))
            return yg.detach(), xg.grad, qsg.grad, ksg.grad
        finally:
            hook(1)
    new, old = run(True), run(False)
    for got in (new, old):
        assert_close(got[0], yo, rtol=3e-2, atol=3e-2, what="out")
        assert_close_scaled(got[1], xo.grad, rel=5e-2, what="dqkv")
        assert_close_scaled(got[2], qso.grad, rel=5e-2, what="dq_scale")
        assert_close_scaled(got[3], kso.grad, rel=5e-2, what="dk_scale")
    # the two implementations agree with each other much more closely than either is required to agree with the oracle
    assert_close(new[0], old[0], rtol=2e-2, atol=2e-2, what="out, matrix-core vs VALU")
    assert_close_scaled(new[1], old[1], rel=3e-2, what="dqkv, matrix-core vs VALU")


@pytest.mark.parametrize("case", [(16, 16, 16, (2, 5, 20, 36)), (16, 16, 32, (1, 4, 16, 32)), (16, 16, 16, (1, 3, 9, 21))])
def test_conv3d_over_two_tensors_equals_conv_of_concat(dev, case):
    """concat([up, skip]) + conv1 of the decoder (reference train/unet.py:79-81) with the two operands left as two dense tensors:
    forward (with the GroupNorm partials), both input gradients, the weight and bias gradients are BITWISE those of the same layer
    run on the concatenated tensor (same kernels, same summation order -- only the addresses differ), and the concatenated run is
    oracle-checked by test_conv3d_bf16_fast_path; through autograd the module path gives the same numbers again."""
    from video_vae_amd import ops
    ca, cb, co, (n, t, h, w) = case
    xa = _bf16_exact((n, t, h, w, ca), 60, 1.0).to(dev, torch.bfloat16)
    xb = _bf16_exact((n, t, h, w, cb), 61, 1.0).to(dev, torch.bfloat16)
    k = _bf16_exact((3, 3, 3, ca + cb, co), 62, (27 * (ca + cb)) ** -0.5).to(dev)
    b = rnd((co,), 63, 0.1).to(dev)
    gy = _bf16_exact((n, t, h, w, co), 64, 1.0).to(dev, torch.bfloat16)
    assert ops.conv3d_cat2_ok(xa, xb, k)
    xc = torch.cat([xa, xb], dim=-1)
    groups = min(8, co)
    nblk = ops.conv3d_gn_blocks(xc, k, groups)
    assert nblk > 0
    y_ref, part_ref = ops.conv3d_fwd_gn_raw(xc, k, b, groups, nblk)
    y, part = ops.conv3d_cat2_fwd_raw(xa, xb, k, b, groups, nblk)
    assert torch.equal(y, y_ref) and torch.equal(part, part_ref)
    assert torch.equal(ops.conv3d_cat2_fwd_raw(xa, xb, k, b), ops.conv3d_fwd_raw(xc, k, b))
    dx_ref = ops.conv3d_dgrad_raw(gy, k)
    dxa, dxb = ops.conv3d_cat2_dgrad_raw(gy, k, ca)
    assert dxa.is_contiguous() and dxb.is_contiguous()
    assert torch.equal(dxa, dx_ref[..., :ca]) and torch.equal(dxb, dx_ref[..., ca:])
    dw_ref, db_ref = ops.conv3d_wgrad_raw(xc, gy, tuple(k.shape))
    dw, db = ops.conv3d_cat2_wgrad_raw(xa, xb, gy, tuple(k.shape))
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    # pitched operands (channel slices of wider buffers) are taken as they are
    wide = torch.zeros((n, t, h, w, ca + 8), dtype=torch.bfloat16, device=dev)
    wide[..., :ca] = xa
    assert torch.equal(ops.conv3d_cat2_fwd_raw(wide[..., :ca], xb, k, b), y_ref)
    # autograd node
    xar, xbr = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
    kr, br = k.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yy, stats = ops.conv3d_cat2_with_gn_stats(xar, xbr, kr, br, groups)
    assert stats is not None and torch.equal(yy, y_ref)
    yy.backward(gy)
    assert torch.equal(xar.grad, dxa) and torch.equal(xbr.grad, dxb) and torch.equal(kr.grad, dw_ref) and torch.equal(br.grad, db_ref)


def test_unet_decoder_two_tensor_level_matches_joint_buffer_free_path(dev):
    """The 16 + 16 channel decoder level of the bf16 UNet (two dense tensors into conv1) against the same network with that level
    forced onto torch.cat: outputs and every parameter gradient agree bitwise."""
    import video_vae_amd as V
    from video_vae_amd import ops, unet as U
    net = V.UNet(12, 16, 2, 3, rngs=V.Rngs(4), dtype=torch.bfloat16).to(dev)
    with torch.no_grad():
        net.final_conv.kernel.copy_(rnd(net.final_conv.kernel.shape, 7, 0.2))
    x = torch.rand(1, 4, 32, 32, 12, generator=torch.Generator().manual_seed(3)).to(dev)
    gy = rnd((1, 4, 32, 32, 3), 5).to(dev, torch.bfloat16)

    def run():
        for p in net.parameters():
            p.grad = None
        y = net(x)
        y.backward(gy)
        return y.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}
    y1, g1 = run()
    real = ops.conv3d_cat2_ok
    ops.conv3d_cat2_ok = lambda *a: False
    try:
        y0, g0 = run()
    finally:
        ops.conv3d_cat2_ok = real
    assert torch.equal(y1, y0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k


def test_patch_mixer_real_channel_product_equals_padded_product(dev):
    """The 3x7x7 mixer on 16-channel voxels of which 12 are real (UNet.forward pads 12 -> 16, reference train/unet.py:111-113): told
    the real count, the rolling kernel multiplies K = 3*7*12 = 252 instead of 336.  Same products in a different order: outputs and
    input gradients agree with the padded product to bf16 rounding of fp32 sums (mostly bitwise), and with the oracle."""
    from video_vae_amd import ops
    n, t, h, w, c, pad = 1, 5, 40, 52, 12, 4
    x = torch.zeros(n, t, h, w, c + pad)
    x[..., :c] = _bf16_exact((n, t, h, w, c), 70, 1.0)
    k = torch.zeros(3, 7, 7, c + pad, c + pad)
    k[..., :c, :c] = _bf16_exact((3, 7, 7, c, c), 71, (147 * c) ** -0.5)
    b = torch.zeros(c + pad); b[:c] = rnd((c,), 72, 0.1)
    gy = torch.zeros(n, t, h, w, c + pad)
    gy[..., :c] = _bf16_exact((n, t, h, w, c), 73, 1.0)
    xg, kg, bg, gyg = x.to(dev, torch.bfloat16), k.to(dev), b.to(dev), gy.to(dev, torch.bfloat16)
    y_pad = ops.conv3d_fwd_raw(xg, kg, bg)
    y_real = ops.conv3d_fwd_raw(xg, kg, bg, k_real=c)
    dx_pad = ops.conv3d_dgrad_raw(gyg, kg)
    dx_real = ops.conv3d_dgrad_raw(gyg, kg, k_real=c)
    assert float(y_real[..., c:].float().abs().max()) == 0 and float(dx_real[..., c:].float().abs().max()) == 0
    assert_close(y_real, y_pad, rtol=1e-2, atol=1e-2, what="real-K vs padded y")
    assert_close(dx_real, dx_pad, rtol=1e-2, atol=1e-2 * float(dx_pad.float().abs().max()), what="real-K vs padded dx")
    assert float((y_real != y_pad).float().mean()) < 0.05 and float((dx_real != dx_pad).float().mean()) < 0.05
    xo = x.clone().requires_grad_(True)
    yo = O.conv3d_same(xo, k, b, torch.bfloat16)
    yo.backward(gy)
    assert_close(y_real, yo, rtol=2e-2, atol=2e-2, what="real-K vs oracle y")
    assert_close_scaled(dx_real, xo.grad, rel=2e-2, what="real-K vs oracle dx")
    # through the prepacked path and autograd, as UNet.forward runs it
    pack = ops.conv3d_prepack([kg], [(c, c)])[0]
    xr = xg.clone().requires_grad_(True)
    yy = ops.conv3d(xr, kg, bg, pack=pack, real=(c, c))
    yy.backward(gyg)
    assert torch.equal(yy, y_real) and torch.equal(xr.grad, dx_real)
    # the per-frame kernel (rolling kernel switched off) ignores the hint: padded product, padded packing
    ops.lib().vvae_conv3d_roll_config(0, 0)
    try:
        assert torch.equal(ops.conv3d_fwd_raw(xg, kg, bg, k_real=c), ops.conv3d_fwd_raw(xg, kg, bg))
    finally:
        ops.lib().vvae_conv3d_roll_config(1, 0)


@pytest.mark.parametrize("m,k,n", [(16384, 512, 768), (4096, 1536, 768), (256, 64, 64)])
def test_linear_residual_library_product(dev, m, k, n):
    """y = x W + bias + res in one library product (vvae_linear_residual_bf16: hipBLASLt with the residual as its C operand) against
    the fp32 oracle of `x_skip + Linear(...)` (reference train/layers.py:151,189,212-221); bf16 operands, one rounding.  Pitched x / res
    rows and the plain (res = None) form too."""
    from video_vae_amd import ops
    x = _bf16_exact((m, k), 80, 1.0)
    w = _bf16_exact((k, n), 81, k ** -0.5)
    b = _bf16_exact((n,), 82, 0.1)
    r = _bf16_exact((m, n), 83, 1.0)
    ref = x.double() @ w.double() + b.double() + r.double()
    xg, wg, bg, rg = (t.to(dev, torch.bfloat16) for t in (x, w, b, r))
    assert ops.linear_residual_ok(xg, wg, bg, rg)
    y = ops.linear_residual(xg, wg, bg, rg)
    assert y.dtype == torch.bfloat16 and y.shape == (m, n)
    assert_close(y, ref, rtol=1e-2, atol=2e-2, what="x W + b + res")
    # not worse than the separate path (product rounded to bf16, then the add rounded again)
    two_step = (torch.addmm(bg, xg, wg) + rg)
    e1 = float((y.double().cpu() - ref).abs().mean()); e2 = float((two_step.double().cpu() - ref).abs().mean())
    assert e1 <= e2 * 1.05 + 1e-6, (e1, e2)
    assert_close(ops.linear_residual(xg, wg, bg, None), x.double() @ w.double() + b.double(), rtol=1e-2, atol=2e-2, what="plain form")
    wide_x = torch.zeros((m, k + 8), dtype=torch.bfloat16, device=dev); wide_x[:, :k] = xg
    wide_r = torch.zeros((m, n + 8), dtype=torch.bfloat16, device=dev); wide_r[:, :n] = rg
    assert torch.equal(ops.linear_residual(wide_x[:, :k], wg, bg, wide_r[:, :n]), y)
    assert torch.equal(ops.linear_residual(xg, wg, bg, rg), y), "bitwise reproducible"
    y_t = ops.linear_residual(xg, wg, bg, rg, wg.t().contiguous())       # the (N, K) shadow form: another library kernel, same product
    assert_close(y_t, ref, rtol=1e-2, atol=2e-2, what="x W + b + res, transposed weight")
    e3 = float((y_t.double().cpu() - ref).abs().mean())
    assert e3 <= e2 * 1.05 + 1e-6, (e3, e2)


def test_silu_stream_kernel(dev):
    """vvae_silu_bf16 (the MLP's activation, reference train/layers.py:186-189) against fp32 silu rounded to bf16: within one bf16 ulp
    everywhere (v_rcp_f32 sigmoid), bitwise on all but a sliver; odd sizes fall back to the framework op."""
    from video_vae_amd import ops
    x = (torch.randn(4099 * 8, generator=torch.Generator().manual_seed(90)) * 3).to(dev, torch.bfloat16)
    y = ops.silu_bf16(x)
    ref = torch.nn.functional.silu(x.float()).to(torch.bfloat16)
    assert_close(y, ref, rtol=2 ** -7, atol=1e-6, what="silu")
    assert float((y != ref).float().mean()) < 0.02
    big = torch.randn(16384, 1536, device=dev, dtype=torch.bfloat16)
    assert_close(ops.silu_bf16(big), torch.nn.functional.silu(big.float()).to(torch.bfloat16), rtol=2 ** -7, atol=1e-6, what="silu, production shape")
    odd = torch.randn(13, device=dev, dtype=torch.bfloat16)
    assert torch.equal(ops.silu_bf16(odd), torch.nn.functional.silu(odd))


def test_transposed_shadows_follow_the_optimizer(dev):
    """The (out, in) bf16 shadows of the kernels marked ``want_t`` (the MLP's fc1) equal the bf16 shadow
    transposed at construction, after every Adam update and after refresh_shadow (vvae_transpose_grouped_bf16)."""
    import video_vae_amd as V
    from video_vae_amd import layers as LY, optim
    m = LY.MLP(128, 320, V.Rngs(2)).to(dev)                        # 320 x 128: both dims multiples of 64, not square
    opt = optim.Optimizer(m, 1e-2)
    k = m.linear1.kernel
    assert k.bf16_t.shape == (320, 128) and torch.equal(m.linear2.kernel.bf16_t, m.linear2.kernel.bf16.t())
    assert getattr(m.norm.scale, "bf16_t", None) is None
    assert torch.equal(k.bf16_t, k.bf16.t())
    x = rnd((256, 128), 5, 1.0).to(dev, torch.bfloat16)
    for _ in range(2):
        opt.zero_grad()
        m(x).float().square().mean().backward()
        before = k.bf16_t.clone()
        opt.update()
        assert torch.equal(k.bf16_t, k.bf16.t()) and not torch.equal(k.bf16_t, before)
    with torch.no_grad():
        k.mul_(0.5)
    opt.refresh_shadow()
    assert torch.equal(k.bf16_t, k.detach().to(torch.bfloat16).t())


@pytest.mark.parametrize("rows,c,mlp", [(4096, 768, 1536), (1024, 512, 1024)])
def test_mlp_residual_on_the_nt_products(dev, rows, c, mlp):
    """MLP.residual with the transposed shadows (fc1 + SiLU out of one NT product, ops.gemm_nt EPI_SILU) against the same block
    with the shadows removed (library product + SiLU stream kernel) and against the fp32 oracle; reference train/layers.py:174-196."""
    import video_vae_amd as V
    from video_vae_amd import layers as LY, optim
    from oracle import layers as OL
    m = LY.MLP(c, mlp, V.Rngs(3)).to(dev)
    opt = optim.Optimizer(m, 1e-3)
    with torch.no_grad():
        m.linear1.bias.copy_(rnd((mlp,), 1, 0.1).to(dev)); m.linear2.kernel.mul_(30.0)
    opt.refresh_shadow()
    x = rnd((rows, c), 7, 1.0).to(dev, torch.bfloat16)
    gy = rnd((rows, c), 8, 1.0).to(dev, torch.bfloat16)
    assert LY.nt_silu_ok(m.linear1, x) and not LY.nt_silu_ok(m.linear1, x[:1000])     # ragged row counts keep the library path
    assert m.residual(x[:1000]).shape == (1000, c)
    out = []
    for fused in (True, False):
        wt = m.linear1.kernel.bf16_t
        if not fused:
            m.linear1.kernel.bf16_t = None
        xx = x.clone().requires_grad_(True)
        opt.zero_grad()
        y = m.residual(xx)
        y.backward(gy)
        for b in range(len(opt.buckets)):
            if not opt.landed[b]:
                opt._land(b)
        out.append((y.detach().float().cpu(), xx.grad.float().cpu(), m.linear1.kernel.gview.clone().cpu(), m.linear1.bias.gview.clone().cpu()))
        m.linear1.kernel.bf16_t = wt
    (y1, dx1, dw1, db1), (y0, dx0, dw0, db0) = out
    p = {n: t.detach().float().cpu() for n, t in m.named_parameters()}
    ref = x.float().cpu() + OL.mlp(p, x.float().cpu())
    e1 = float((y1 - ref).abs().mean()); e0 = float((y0 - ref).abs().mean())
    assert e1 <= 1.1 * e0 + 1e-4, (e1, e0)
    assert_close_scaled(y1, y0, rel=1e-2, what="y nt vs library")
    assert_close_scaled(dx1, dx0, rel=1e-2, what="dx")
    assert_close_scaled(dw1, dw0, rel=1e-2, what="dW1")
    assert_close_scaled(db1, db0, rel=1e-2, what="db1")


@pytest.mark.parametrize("b,t", [(4, 16), (3, 5), (64, 32)])
def test_plain_loss_tail_matches_the_framework_ops(dev, b, t):
    """vvae_loss_tail_plain (loss, aux and all three gradients in one launch) against the same algebra as differentiable framework
    ops — the reference's per-sample tail, train/legacy/training_loop_adversarial.py:100-124; ragged masks, an all-masked sample
    (length clamps to 1), densities on both sides of 1 / max_compression_rate."""
    from video_vae_amd import ops
    from video_vae_amd.loss import HPARAMS, magnify_negatives
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand(b, t, generator=g) < 0.7).float()
    mask[0] = 0.0
    mask[-1] = 1.0
    sel = torch.rand(b, t, 1, 1, generator=g)
    sel[1] = torch.round(sel[1])
    mse = torch.rand(b, generator=g); kl = torch.rand(b, generator=g) * 50
    leaves = [x.to(dev).requires_grad_(True) for x in (mse, kl, sel)]
    mk = mask.to(dev)
    assert ops.plain_loss_tail_ok(leaves[0], leaves[1], leaves[2], mk)
    loss, (MSE, sl, klm, dens) = ops.plain_loss_tail(leaves[0], leaves[1], leaves[2], mk, HPARAMS)
    (loss * 3.0).backward()
    got = [x.grad.clone() for x in leaves]
    ref_leaves = [x.double().requires_grad_(True) for x in (mse, kl, sel)]
    m64 = mask.double()
    lens = torch.clamp(m64.sum(1, keepdim=True), min=1.0)
    density = (ref_leaves[2].reshape(b, t) * m64).sum(1, keepdim=True) / lens
    s_loss = torch.square(magnify_negatives(density - 1 / HPARAMS["max_compression_rate"], HPARAMS["magnify_negatives_rate"])).mean()
    ref = ref_leaves[0].mean() + HPARAMS["gamma1"] * s_loss + HPARAMS["gamma2"] * ref_leaves[1].mean()
    (ref * 3.0).backward()
    for a, w, what in ((loss, ref, "loss"), (MSE, ref_leaves[0].mean(), "MSE"), (sl, s_loss, "selection_loss"),
                       (klm, ref_leaves[1].mean(), "kl"), (dens, density.mean(), "density")):
        assert_close(a, w.detach(), rtol=1e-5, atol=1e-6, what=what)
    for a, w, what in zip(got, ref_leaves, ("d mse", "d kl", "d selection")):
        assert a.shape == w.grad.shape
        assert_close_scaled(a, w.grad, rel=1e-5, what=what)      # fp32 density - 1/rate against the fp64 reference: absolute, not relative


def test_plain_loss_tail_sums_kl_partials(dev):
    """kl_ps as (b, k) partial sums of the per-sample term (one per frame from ops.encoder_head): same loss and kl_loss as their row
    sums, and every partial receives its sample's gradient."""
    from video_vae_amd import ops
    from video_vae_amd.loss import HPARAMS
    b, t, k = 3, 6, 6
    g = torch.Generator().manual_seed(9)
    mask = (torch.rand(b, t, generator=g) < 0.8).float().to(dev)
    sel = torch.rand(b, t, 1, 1, generator=g).to(dev)
    mse = torch.rand(b, generator=g).to(dev)
    klp = (torch.rand(b, k, generator=g) * 10).to(dev).requires_grad_(True)
    kls = klp.detach().sum(1).requires_grad_(True)
    assert ops.plain_loss_tail_ok(mse, klp, sel, mask)
    l2, aux2 = ops.plain_loss_tail(mse, klp, sel, mask, HPARAMS)
    l1, aux1 = ops.plain_loss_tail(mse, kls, sel, mask, HPARAMS)
    l2.backward(); l1.backward()
    assert_close(l2, l1.detach(), rtol=1e-6, atol=1e-7, what="loss")
    assert_close(aux2[2], aux1[2].detach(), rtol=1e-6, atol=1e-7, what="kl_loss")
    assert klp.grad.shape == (b, k)
    assert torch.equal(klp.grad, kls.grad[:, None].expand(b, k))
    # the MSE term likewise: the per-workgroup partials of ops.masked_mse_mae(partials=True) against their row sums, through to d recon
    bf = torch.bfloat16
    video = torch.rand(b, t, 8, 8, 3, generator=g).to(dev, bf)
    outs = []
    for partials in (True, False):
        recon = torch.rand(b, t, 8, 8, 3, generator=torch.Generator().manual_seed(11)).to(dev, bf).requires_grad_(True)
        mse, mae = ops.masked_mse_mae(video, recon, mask, 1, partials)
        assert mse.dim() == (2 if partials else 1)
        loss, _ = ops.plain_loss_tail(mse, kls.detach(), sel, mask, HPARAMS)
        loss.backward()
        outs.append((loss.detach(), mse.detach().reshape(b, -1).sum(1), recon.grad))
    assert_close(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-7, what="loss from MSE partials")
    assert_close(outs[0][1], outs[1][1], rtol=1e-6, atol=1e-7, what="per-sample MSE")
    assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("b,t", [(4, 16), (1, 5), (8, 32)])
def test_rl_loss_tail_matches_the_framework_ops(dev, b, t):
    """vvae_loss_tail_rl (the pair / REINFORCE end of loss.loss_fn, value + all gradients in one launch) against the same algebra as
    framework ops (loss.rl_loss_tail_ops, reference train/rl_nonadversarial.py:130-186): ragged masks, an all-masked sample, probabilities at
    and beyond the clip bounds, MSE / MAE as partial sums."""
    from video_vae_amd import ops
    from video_vae_amd.loss import HPARAMS, rl_loss_tail_ops
    g = torch.Generator().manual_seed(b * 100 + t)
    b2 = 2 * b
    mask = (torch.rand(b2, t, generator=g) < 0.8).float()
    mask[0] = 0.0
    mask[-1] = 1.0
    sel = torch.rand(b2, t, 1, 1, generator=g)
    sel.view(-1)[1] = 0.0; sel.view(-1)[2] = 1.0                       # the clip bounds: no gradient there
    act = (torch.rand(b2, t, 1, 1, generator=g) < sel).float()
    cols = 3
    mse_p = torch.rand(b2, cols, generator=g); mae_p = torch.rand(b2, cols, generator=g)
    kl = torch.rand(b2, generator=g) * 20
    leaves = [x.to(dev).requires_grad_(True) for x in (mse_p, mae_p, kl, sel)]
    mk, ak = mask.to(dev), act.to(dev)
    assert ops.rl_loss_tail_ok(leaves[0], leaves[1], leaves[2], leaves[3], ak, mk)
    loss, aux = ops.rl_loss_tail(leaves[0], leaves[1], None, leaves[2], leaves[3], ak, mk, HPARAMS)
    (loss * 2.0).backward()
    got = [x.grad.clone() for x in leaves]
    ref = [x.double().requires_grad_(True) for x in (mse_p, mae_p, kl, sel)]
    mse_r, mae_r = ref[0].sum(1), ref[1].sum(1)
    rl, ra = rl_loss_tail_ops(mse_r, mae_r, torch.zeros_like(mse_r), ref[2], ref[3], act.double(), mask.double(), HPARAMS)
    (rl * 2.0).backward()
    names = ("MSE", "perceptual_loss", "selection_loss", "kl_loss", "kept_frame_density", "mean_trajectory_prob", "rl_loss", "per_sample_MAE")
    assert_close(loss, rl.detach(), rtol=1e-4, atol=1e-5, what="loss")
    for a, n in zip(aux, names):
        assert_close(a, ra[n].detach(), rtol=1e-4, atol=1e-5, what=n)
    for a, w, what in zip(got, ref, ("d mse", "d mae", "d kl", "d selection")):
        assert a.shape == w.grad.shape
        assert_close_scaled(a, w.grad, rel=1e-4, what=what)


@pytest.mark.parametrize("b,t,hw,ld", [(2, 3, 16, 96), (1, 4, 256, 96), (3, 2, 10, 8), (2, 2, 4, 64)])
def test_rl_gate_matches_the_framework_ops(dev, b, t, hw, ld):
    """ops.rl_gate (pair doubling + Bernoulli frame masks + fill (1 - mask) + z mask of the rl flavour, one launch each way) against the
    framework ops of rl_model.VideoVAE.forward: the same mask, the same values rounded to the decoder's dtype, the same gradients."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(b * 7 + hw)
    z = torch.randn(b, t, hw, ld, generator=g).to(dev)
    prob = torch.rand(b, t, 1, generator=g).to(dev)
    u = torch.rand(2 * b, t, 1, 1, generator=g).to(dev)
    fill = (torch.randn(1, 1, 1, ld, generator=g) * 0.02).to(dev)
    gc = torch.randn(2 * b, t, hw, ld, generator=g).to(dev, torch.bfloat16)
    assert ops.rl_gate_ok(z, prob, fill)
    z1, f1 = z.clone().requires_grad_(True), fill.clone().requires_grad_(True)
    comp1, m1 = ops.rl_gate(z1, prob, u, f1)
    g1 = torch.autograd.grad(comp1, [z1, f1], gc)
    z0, f0 = z.clone().requires_grad_(True), fill.clone().requires_grad_(True)
    sel = prob[..., None].repeat_interleave(2, dim=0)
    m0 = (u < sel).to(torch.float32)
    comp0 = (f0 * (1 - m0) + z0.repeat_interleave(2, dim=0) * m0).to(torch.bfloat16)
    g0 = torch.autograd.grad(comp0, [z0, f0], gc)
    assert comp1.dtype == torch.bfloat16 and m1.shape == (2 * b, t, 1, 1) and torch.equal(m1, m0)
    assert 0 < float(m1.sum()) < 2 * b * t
    assert torch.equal(comp1, comp0)
    assert_close_scaled(g1[0], g0[0], rel=1e-6, what="dz")
    assert_close_scaled(g1[1], g0[1], rel=1e-4, what="d fill")


def _heads_reference(mean, v, w1, b1, w2, b2, fill, u, eps, mask_bt):
    """The unfused framework path of model.Encoder._trunk / GumbelSigmoidSTE / VideoVAE.forward on the same leaves (bf16 compute)."""
    import torch.nn.functional as F
    from video_vae_amd import ops
    from video_vae_amd.layers import round_ste
    bf = torch.bfloat16
    lv = torch.log(F.softplus(v))
    b, t, hw, ld = mean.shape
    s1 = torch.addmm(b1.to(bf), mean.reshape(-1, ld), w1.to(bf)).view(b, t, hw)          # layers._LinearBf16.forward
    logits = torch.addmm(b2.to(bf), s1.reshape(-1, hw), w2.to(bf)).view(b, t, 1) + 1
    y = logits.float() + torch.logit(u, eps=1e-20)
    sel = round_ste(torch.sigmoid(y))[..., None]                    # (b, t, 1, 1)
    z, kl = ops.reparameterise_kl(mean, lv, eps, mask_bt)
    comp = fill * (1 - sel) + z * sel
    return lv, comp, sel, kl


@pytest.mark.parametrize("b,t,hw,ld", [(2, 3, 16, 96), (1, 4, 256, 96), (2, 2, 40, 8), (1, 2, 100, 200)])
def test_encoder_head_matches_the_framework_ops(dev, b, t, hw, ld):
    """ops.encoder_head (the model.py flavour's heads + Gumbel STE + reparameterisation + KL + latent gate, one launch each way) against the
    framework ops it replaces, on the same leaves: identical selection, outputs and every gradient within bf16 rounding.  Frames on both
    sides of the gate, a masked frame, gradients arriving at comp, selection, the KL term and log_variance."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(100 + hw)
    bf = torch.bfloat16
    mean = (torch.randn(b, t, hw, ld, generator=g) * 0.5).to(dev, bf)
    v = (torch.randn(b, t, hw, ld, generator=g) * 1.5).to(dev, bf)
    w1 = (torch.randn(ld, 1, generator=g) * ld ** -0.5).to(dev)
    b1 = (torch.randn(1, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(hw, 1, generator=g) * hw ** -0.5).to(dev)
    b2 = (torch.randn(1, generator=g) * 0.1).to(dev)
    fill = (torch.randn(1, 1, 1, ld, generator=g) * 0.02).to(dev)
    # Gumbel noise well away from the decision boundary (logits are 1 + O(0.3)): u < 0.1 drops the frame, u > 0.9 keeps it, so that a
    # last-bit difference in the bf16 logits cannot flip a frame between the two paths
    r = torch.rand(b, t, 1, generator=g)
    u = torch.where(torch.rand(b, t, 1, generator=g) < 0.5, 0.02 + 0.08 * r, 0.9 + 0.08 * r)
    u.view(-1)[0], u.view(-1)[-1] = 0.02, 0.98
    u = u.to(dev)
    eps = torch.randn(b, t, hw, ld, generator=g).to(dev)
    mask = torch.ones(b, t); mask[0, -1] = 0.0
    mask_x = mask[:, None, :].expand(b, hw, t).reshape(b * hw, 1, 1, t).to(dev)          # as train_step expands it
    from video_vae_amd.model import frame_mask
    mbt = frame_mask(mask_x, b, t)
    gcomp = torch.randn(b, t, hw, ld, generator=g).to(dev, bf)
    gsel = torch.randn(b, t, 1, 1, generator=g).to(dev)
    gkl = torch.randn(b, generator=g).to(dev)
    glv = (torch.randn(b, t, hw, ld, generator=g) * 0.1).to(dev, bf)
    assert ops.encoder_head_ok(mean, v, w1, b1, w2, b2, fill)

    def run(fused):
        leaves = [x.clone().requires_grad_(True) for x in (mean, v, w1, b1, w2, b2, fill)]
        if fused:
            lv, comp, sel, kl = ops.encoder_head(*leaves, u, eps, mbt)
            kl = kl.sum(1)
        else:
            lv, comp, sel, kl = _heads_reference(*leaves, u, eps, mbt)
        tot = (comp.float() * gcomp.float()).sum() + (sel * gsel).sum() + (kl * gkl).sum() + (lv.float() * glv.float()).sum()
        tot.backward()
        return (lv, comp, sel, kl), [x.grad for x in leaves]

    (lv1, c1, s1, k1), g1 = run(True)
    (lv0, c0, s0, k0), g0 = run(False)
    assert c1.dtype == bf and lv1.dtype == bf and s1.shape == (b, t, 1, 1) and s1.dtype == torch.float32
    assert torch.equal(s1, s0), (s1.flatten(), s0.flatten())
    assert 0 < float(s1.detach().sum()) < b * t                    # both sides of the gate are exercised
    assert_close(lv1.float(), lv0.float(), rtol=8e-3, atol=1e-6, what="log_variance")      # the same roundings: at most a bf16 ulp apart
    assert_close_scaled(c1.float(), c0.float(), rel=1e-2, what="comp")
    assert_close(k1, k0.detach(), rtol=1e-4, atol=1e-6, what="kl")
    for a, w, what in zip(g1, g0, ("d mean", "d v", "d w1", "d b1", "d w2", "d b2", "d fill")):
        assert a is not None and a.shape == w.shape and a.dtype == w.dtype, what
        assert_close_scaled(a.float(), w.float(), rel=2e-2, what=what)


@pytest.mark.parametrize("v,c,dt", [(16384, 768, torch.bfloat16), (16384, 96, torch.bfloat16), (1000, 8, torch.bfloat16), (777, 20, torch.float32),
                                    (4096, 2048, torch.bfloat16), (300, 7, torch.bfloat16), (64, 1, torch.bfloat16)])
def test_colsum_vector_and_scalar_forms(dev, v, c, dt):
    """vvae_colsum (bias gradients of the Linear layers outside the GEMM kernels' shapes): the 16-byte vector form, the column-group loop
    (more than 256 column vectors), the scalar fallback (odd widths), a pitched slice; against the fp64 column sums, twice (bitwise equal)."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(v + c)
    x = torch.randn(v, c + 8, generator=g).to(dev, dt)[:, :c] if c % 8 == 0 else torch.randn(v, c, generator=g).to(dev, dt)
    a = ops.colsum_raw(x)
    b = ops.colsum_raw(x)
    assert torch.equal(a, b) and a.shape == (c,) and a.dtype == torch.float32
    assert_close_scaled(a, x.double().sum(0), rel=2e-5 if dt == torch.float32 else 1e-4, what="colsum")


@pytest.mark.parametrize("c,h,w,dt,sliced", [(16, 32, 32, torch.bfloat16, False), (32, 16, 24, torch.bfloat16, True), (64, 8, 8, torch.bfloat16, True),
                                              (128, 4, 6, torch.bfloat16, False), (16, 8, 8, torch.float32, False)])
def test_group_norm_silu_with_pool(dev, c, h, w, dt, sliced):
    """ops.group_norm_silu(pool=True) (GroupNorm + SiLU + the (1,2,2) max-pool of an encoder level in one launch, vvae_gn_silu_pool_fwd)
    against GroupNorm + SiLU followed by ops.max_pool_fork: bitwise the same outputs -- also into the channel half of a joint buffer -- and,
    the backward being the same kernels in the same order, bitwise the same gradients."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(c + h)
    groups = min(8, c)
    x = torch.randn(2, 3, h, w, c, generator=g).to(dev, dt)
    sc = (1 + 0.1 * torch.randn(c, generator=g)).to(dev)
    bi = (0.1 * torch.randn(c, generator=g)).to(dev)
    gy = torch.randn(2, 3, h, w, c, generator=g).to(dev, dt)
    gp = torch.randn(2, 3, h // 2, w // 2, c, generator=g).to(dev, dt)
    res = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (x, sc, bi)]
        out = torch.zeros(2, 3, h, w, 2 * c, device=dev, dtype=dt)[..., c:] if sliced else None
        assert ops.gn_silu_pool_ok(leaves[0], groups, out)
        if fused:
            y, p = ops.group_norm_silu(leaves[0], leaves[1], leaves[2], groups, 1e-6, out, pool=True)
        else:
            p, y = ops.max_pool_fork(ops.group_norm_silu(leaves[0], leaves[1], leaves[2], groups, 1e-6, out))
        grads = torch.autograd.grad([y, p], leaves, [gy, gp])
        res.append((y.detach().clone(), p.detach().clone(), grads))
    (y1, p1, g1), (y0, p0, g0) = res
    assert torch.equal(y1, y0) and torch.equal(p1, p0)
    assert torch.equal(p1, torch.nn.functional.max_pool3d(y1.permute(0, 4, 1, 2, 3).float(), (1, 2, 2)).permute(0, 2, 3, 4, 1).to(dt))
    for a, b, what in zip(g1, g0, ("dx", "dscale", "dbias")):           # the backward is the same kernels in the same order
        assert torch.equal(a, b), what


def test_pad_last2_group_matches_f_pad(dev):
    """ops.pad_last2_group (the UNet's three 12-channel weight pads in one launch each way) against F.pad: values, and the gradients
    of a padded-space cotangent cut back to the parameters' shapes."""
    import torch.nn.functional as F
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(3)
    km = torch.randn(3, 7, 7, 12, 12, generator=g).to(dev).requires_grad_(True)
    bm = torch.randn(12, generator=g).to(dev).requires_grad_(True)
    k1 = torch.randn(3, 3, 3, 12, 16, generator=g).to(dev).requires_grad_(True)
    kd = torch.randn(12, 3, generator=g).to(dev).requires_grad_(True)
    outs = ops.pad_last2_group([km, bm, k1, kd], [(16, 16), (16,), (16, 16), (16, 3)])
    want = [F.pad(km, (0, 4, 0, 4)), F.pad(bm, (0, 4)), F.pad(k1, (0, 0, 0, 4)), F.pad(kd, (0, 0, 0, 4))]
    cots = [torch.randn(w.shape, generator=g).to(dev) for w in want]
    for o, w in zip(outs, want):
        assert o.shape == w.shape and torch.equal(o, w.detach())
    grads = torch.autograd.grad(outs, [km, bm, k1, kd], cots)
    refs = torch.autograd.grad(want, [km, bm, k1, kd], cots)
    for a, r in zip(grads, refs):
        assert a.shape == r.shape and torch.equal(a, r)


def test_pointwise_conv_with_addend(dev):
    """ops.conv3d_pointwise_add = addend + conv1x1x1(x) (the decoder's coarse + UNet(features), reference train/model.py:97) in one
    launch: against the two-launch form, forward (one rounding fewer) and all four gradients."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(4)
    bf = torch.bfloat16
    x = torch.randn(2, 3, 16, 24, 16, generator=g).to(dev, bf).requires_grad_(True)
    add = torch.randn(2, 3, 16, 24, 3, generator=g).to(dev, bf).requires_grad_(True)
    k = (torch.randn(1, 1, 1, 16, 3, generator=g) * 0.3).to(dev).requires_grad_(True)
    b = torch.randn(3, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(2, 3, 16, 24, 3, generator=g).to(dev, bf)
    assert ops.conv3d_pointwise_add_ok(x, k, add)
    y1 = ops.conv3d_pointwise_add(x, k, b, add)
    g1 = torch.autograd.grad(y1, [x, k, b, add], gy)
    y0 = add + ops.conv3d(x, k, b)
    g0 = torch.autograd.grad(y0, [x, k, b, add], gy)
    ref = add.float() + torch.einsum("...i,io->...o", x.float(), k[0, 0, 0]) + b
    assert float((y1.float() - ref).abs().max()) <= float((y0.float() - ref).abs().max()) + 1e-6
    assert_close_scaled(y1.float(), ref, rel=8e-3, what="y")
    for a, w, what in zip(g1, g0, ("dx", "dk", "db", "d addend")):
        assert a.shape == w.shape and a.dtype == w.dtype
        assert_close_scaled(a.float(), w.float(), rel=1e-5, what=what)


def test_pointwise_conv_fork_adds_the_other_gradient(dev):
    """ops.conv3d_pointwise_fork: (conv1x1x1(x), x) from one node; the gradient of x's other consumer is added inside the projection's
    input-gradient launch.  Against the plain conv + autograd's own accumulation."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(12)
    bf = torch.bfloat16
    x = torch.randn(2, 3, 16, 24, 16, generator=g).to(dev, bf)
    k = (torch.randn(1, 1, 1, 16, 3, generator=g) * 0.3).to(dev).requires_grad_(True)
    b = torch.randn(3, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(2, 3, 16, 24, 3, generator=g).to(dev, bf)
    go = torch.randn(2, 3, 16, 24, 16, generator=g).to(dev, bf)
    x1 = x.clone().requires_grad_(True)
    assert ops.conv3d_pointwise_fork_ok(x1, k)
    y1, xa = ops.conv3d_pointwise_fork(x1, k, b)
    g1 = torch.autograd.grad([y1, xa * 1.0], [x1, k, b], [gy, go])
    x0 = x.clone().requires_grad_(True)
    y0 = ops.conv3d(x0, k, b)
    g0 = torch.autograd.grad([y0, x0 * 1.0], [x0, k, b], [gy, go])
    assert torch.equal(y1, y0) and torch.equal(xa, x)
    assert torch.equal(g1[0], g0[0])                                # round(dy W^T) + other, rounded: the same two roundings
    assert torch.equal(g1[1], g0[1]) and torch.equal(g1[2], g0[2])
    # only one consumer's gradient arrives: the other side is None
    (gx,) = torch.autograd.grad(ops.conv3d_pointwise_fork(x1, k, b)[0], x1, gy)
    (gx0,) = torch.autograd.grad(ops.conv3d(x0, k, b), x0, gy)
    assert torch.equal(gx, gx0)


def test_conv_transpose_prepacked_weights(dev):
    """ops.convt_prepack (both packings of every up-conv kernel in one launch, once per step) gives bitwise the results of the per-call
    packing, forward and input gradient, at the three decoder shapes."""
    from video_vae_amd import ops
    g = torch.Generator().manual_seed(6)
    ks = [(torch.randn(1, 2, 2, ci, co, generator=g) * 0.1).to(dev) for ci, co in ((128, 64), (64, 32), (32, 16))]
    packs = ops.convt_prepack(ks)
    assert all(p is not None for p in packs)
    for k, pk in zip(ks, packs):
        ci, co = k.shape[-2:]
        x = torch.randn(2, 3, 8, 8, ci, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
        b = torch.randn(co, generator=g).to(dev)
        gy = torch.randn(2, 3, 16, 16, co, generator=g).to(dev, torch.bfloat16)
        y1 = ops.conv_transpose_1x2x2(x, k, b, pack=pk)
        y0 = ops.conv_transpose_1x2x2(x, k, b)
        assert torch.equal(y1, y0)
        (d1,) = torch.autograd.grad(y1, x, gy)
        (d0,) = torch.autograd.grad(y0, x, gy)
        assert torch.equal(d1, d0)


def test_linear_pair_matches_two_linears(dev):
    """layers.linear_pair (the encoder's mean / variance heads as one autograd node: the second input-gradient product accumulates onto
    the first) against two Linear calls."""
    import video_vae_amd as V
    from video_vae_amd.layers import Linear, linear_pair
    l1 = Linear(768, 96, V.Rngs(1)).to(dev)
    l2 = Linear(768, 96, V.Rngs(2)).to(dev)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 4, 64, 768, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    g1 = torch.randn(2, 4, 64, 96, generator=g).to(dev, torch.bfloat16)
    g2 = torch.randn(2, 4, 64, 96, generator=g).to(dev, torch.bfloat16)
    prm = [l1.kernel, l1.bias, l2.kernel, l2.bias]
    a1, a2 = linear_pair(x, l1, l2)
    ga = torch.autograd.grad([a1, a2], [x] + prm, [g1, g2])
    b1, b2 = l1(x), l2(x)
    gb = torch.autograd.grad([b1, b2], [x] + prm, [g1, g2])
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert_close_scaled(ga[0].float(), gb[0].float(), rel=8e-3, what="dx")        # one bf16 rounding instead of three
    for a, w in zip(ga[1:], gb[1:]):
        assert torch.equal(a, w)


def test_fp32_fallback_paths_are_bitwise_reproducible(dev):
    """Round 3: the fp32 / odd-shape fallbacks (generic Conv3d and ConvTranspose weight + bias gradients, the any-head-dim temporal
    attention's q/k-norm scale gradients) accumulated with float atomics -- configs C1-C2 were reproducible to rounding only.  They now
    write per-workgroup partials folded in index order: two passes over the same inputs are bitwise equal (and still match the oracle:
    test_conv3d_fwd_bwd / test_conv_transpose / test_temporal_attention cover the values)."""
    import video_vae_amd as V
    from video_vae_amd import ops
    from oracle import unet as OU
    p = OU.init_unet(12, 16, 2, 3, seed=1, zero_final=False)
    m = V.UNet(12, 16, 2, 3, V.Rngs(0), dtype=torch.float32)
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in p.items():
            sd[k].copy_(v)
    m = m.to(dev)
    x = rnd((2, 5, 24, 40, 12), 90, 0.5).to(dev)
    gy = rnd((2, 5, 24, 40, 3), 91).to(dev)
    runs = []
    for _ in range(2):
        for prm in m.parameters():
            prm.grad = None
        m(x).backward(gy)
        runs.append({k: prm.grad.clone() for k, prm in m.named_parameters()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), f"fp32 UNet gradient of {k} differs between two passes"
    # any-head-dim temporal attention (head_dim 24 is not one of the fast kernels' 8/16/32/64)
    heads, d, t, a = 2, 24, 7, 37
    qkv = rnd((a, t, 3 * heads * d), 92).to(dev).requires_grad_(True)
    qs, ks = (1 + 0.1 * rnd((d,), 93)).to(dev).requires_grad_(True), (1 + 0.1 * rnd((d,), 94)).to(dev).requires_grad_(True)
    pos = torch.arange(t, dtype=torch.float32)[:, None] / (10000.0 ** (torch.arange(0, d, 2).float() / d))[None, :]
    cos, sin = torch.cat([pos.cos(), pos.cos()], -1).to(dev), torch.cat([pos.sin(), pos.sin()], -1).to(dev)
    go = rnd((a, t, heads * d), 95).to(dev)
    grads = []
    for _ in range(2):
        for v in (qkv, qs, ks):
            v.grad = None
        ops.temporal_attention_core(qkv, qs, ks, cos, sin, None, 1, heads).backward(go)
        grads.append((qs.grad.clone(), ks.grad.clone(), qkv.grad.clone()))
    for g0, g1 in zip(*grads):
        assert torch.equal(g0, g1)
