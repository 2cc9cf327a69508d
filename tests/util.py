"""Shared helpers for the parity tests (tests only)."""
import torch

RTOL, ATOL = 1e-3, 1e-4          # fp32 parity bar of BASELINE.json:north_star


def assert_close(got, want, rtol=RTOL, atol=ATOL, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any():
        i = (err - tol).argmax()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} outside rtol={rtol} atol={atol}; "
                             f"worst |err|={float(err.flatten()[i]):.3e} at want={float(want.flatten()[i]):.3e}, "
                             f"max|want|={float(want.abs().max()):.3e}")


def assert_close_scaled(got, want, rel=1e-3, what="", floor=0.0):
    """Tolerance relative to the tensor's scale: for gradients whose magnitude is far from 1.

    ``floor``: a lower bound for that scale, for gradients that are exactly zero in exact arithmetic and pure
    rounding noise in floating point (a conv bias in front of a GroupNorm with one channel per group).
    """
    want_c = want.detach().float().cpu()
    scale = max(float(want_c.abs().max()), floor)
    assert_close(got, want, rtol=rel, atol=rel * max(scale, 1e-30), what=what)


def grad_floor(name, ref_grads):
    """Scale floor for the gradient of parameter ``name``: the kernel-gradient scale of the same conv for its bias."""
    if name.endswith("conv.bias"):
        return float(ref_grads[name[:-4] + "kernel"].abs().max())
    return 0.0


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale
