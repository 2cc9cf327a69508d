"""Config C1 of BASELINE.json ("toy 2-layer Conv3d encoder/decoder, B=1 3x8x64x64 random clips") on the CPU.  Test infrastructure only.

The reference has no such model: BASELINE.json names a ``train/toy.py`` path that does not exist (the only ``toy.py`` is
claude_distributed/toy.py:1-36, a sharding demo without a conv).  SURVEY.md R2 therefore defines the config from the reference's own
building block: two ``ConvBlock3D`` (Conv 3x3x3 SAME -> GroupNorm(min(8, C)) -> SiLU, train/unet.py:7-30) as encoder, two as decoder,
with the reparameterisation of train/model.py:124-128 between them and the masked recon + KL loss of
train/legacy/training_loop_adversarial.py:97-102,119-122 on top.  Every piece below is the oracle function that already restates that
reference line; nothing here has semantics of its own beyond the wiring:

    h        = ConvBlock3D(c -> f)(x); h = ConvBlock3D(f -> 2 l)(h)          encoder
    mean, lv = h[..., :l], h[..., l:]                                        latent heads = channel halves
    z        = mean + eps * exp(lv / 2)                                      model.py:124-128
    recon    = ConvBlock3D(f -> c)(ConvBlock3D(l -> f)(z))                   decoder
    loss     = mean_b MSE_b + kl_weight * mean_b KL_b                        training_loop_adversarial.py:97-102,119-122
"""
import torch

from . import loss as OLoss
from . import model as OM
from . import nn as O
from . import unet as OU

KL_WEIGHT = 0.05


def init_toy(channels=3, features=16, latent=8, seed=0, temporal_kernel=3):
    """Parameter tree with Flax default inits (lecun_normal kernels, zero biases, unit GroupNorm scales: unet.py:13-23)."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    OU.init_conv_block(p, "enc1", channels, features, 3, temporal_kernel, gen)
    OU.init_conv_block(p, "enc2", features, 2 * latent, 3, temporal_kernel, gen)
    OU.init_conv_block(p, "dec1", latent, features, 3, temporal_kernel, gen)
    OU.init_conv_block(p, "dec2", features, channels, 3, temporal_kernel, gen)
    return p


def toy_vae(p, x, eps, train=True, dtype=O.F32):
    """-> (reconstruction, z, log_variance, mean), all (b, t, h, w, .)."""
    h = OU.conv_block3d(OU.sub(p, "enc1"), O.q(x, dtype), dtype)
    h = OU.conv_block3d(OU.sub(p, "enc2"), h, dtype)
    l = h.shape[-1] // 2
    mean, logvar = h[..., :l], h[..., l:]
    z = OM.reparameterise(mean, logvar, eps, train, dtype)
    r = OU.conv_block3d(OU.sub(p, "dec1"), z, dtype)
    r = OU.conv_block3d(OU.sub(p, "dec2"), r, dtype)
    return r, z, logvar, mean


def toy_loss(outputs, video, mask_bt, kl_weight=KL_WEIGHT, dtype=O.F32):
    """Masked MSE + kl_weight * KL, both per sample then averaged -> (loss, aux)."""
    recon, _z, logvar, mean = outputs
    om = mask_bt.to(torch.float32)
    mse, _ = OLoss.masked_mse_mae(video, recon, om, dtype)
    b, t = mean.shape[:2]
    kl = OLoss.kl_per_sample(mean.reshape(b, t, -1, mean.shape[-1]), logvar.reshape(b, t, -1, mean.shape[-1]), om, dtype)
    loss = mse.mean() + kl_weight * kl.mean()
    return loss, {"MSE": mse.mean(), "kl_loss": kl.mean(), "reconstruction": recon}
