"""CPU oracle for the video-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the reference's
JAX/Flax algorithm for the path named by BASELINE.json:north_star.  It exists
to *check* the HIP product path; it is never the thing measured or shipped.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package ``video_vae_amd``
never imports it and has no CPU fallback.

PARITY UNPINNED: the reference (floatingtrees/video-VAE) is JAX + Flax NNX;
jax / flax / optax are not installed in this image and there is no network,
so the reference cannot be executed, and its own tests assert only shapes,
finiteness and "loss decreases" -- they hold no golden vectors.  The Flax /
JAX / optax semantics restated here come from library
knowledge; each silent trap (GroupNorm eps=1e-6 + fast variance,
ConvTranspose kernel un-flipped, SAME padding, (in,out) Linear kernels) is
guarded by a self-consistency test in ``tests/test_oracle.py``.

Every function cites the reference file:line it follows
(paths relative to the reference checkout).
"""
from . import nn, unet, layers, model, loss, optim, perceptual  # noqa: F401
