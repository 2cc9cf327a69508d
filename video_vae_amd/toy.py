"""Config C1 of BASELINE.json: the toy Conv3d encoder / decoder VAE ("train/toy.py path"), on the HIP kernels.

BASELINE.json names a ``train/toy.py`` that the reference does not have; the config is defined from the reference's own
block instead: two ``ConvBlock3D`` (train/unet.py:7-30) as encoder, two as decoder, the reparameterisation of train/model.py:124-128
between them, masked recon + KL loss (train/legacy/training_loop_adversarial.py:97-102,119-122).  Every launch is the product's own
(``ops.conv3d`` -> GroupNorm + SiLU, ``ops.reparameterise_kl``, ``ops.masked_mse_mae``); on 3-channel clips the convs take the exact-fp32
matrix-core path (csrc/conv3d_generic.hip).  The CPU counterpart is oracle/toy.py (tests only).
"""
import torch
from torch import nn

from . import ops
from .unet import ConvBlock3D

KL_WEIGHT = 0.05


class ToyVAE(nn.Module):
    """ConvBlock3D(c -> f) -> ConvBlock3D(f -> 2 l) | mean, log_variance = channel halves | z = mean + eps exp(lv / 2) |
    ConvBlock3D(l -> f) -> ConvBlock3D(f -> c).  ``forward(x (b,t,h,w,c), mask (b,t), rngs)`` -> (reconstruction, z, log_variance, mean)."""

    def __init__(self, channels=3, features=16, latent=8, rngs=None, temporal_kernel=3, dtype=torch.float32, param_dtype=torch.float32):
        super().__init__()
        self.enc1 = ConvBlock3D(channels, features, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.enc2 = ConvBlock3D(features, 2 * latent, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.dec1 = ConvBlock3D(latent, features, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.dec2 = ConvBlock3D(features, channels, 3, rngs, temporal_kernel, dtype, param_dtype)
        self.latent, self.dtype = latent, dtype

    def forward(self, x, mask, rngs, train=True):
        self._kl = None
        h = self.enc2(self.enc1(x.to(self.dtype)))
        mean, log_variance = h[..., :self.latent].contiguous(), h[..., self.latent:].contiguous()
        if train:
            eps = rngs.draw("reparam_eps", "normal", mean.shape, mean.device)
            b, t = mean.shape[:2]
            z, kl = ops.reparameterise_kl(mean, log_variance, eps, mask.reshape(b, t))   # z and the per-sample KL term in one pass
            self._kl = kl
        else:
            z = mean
        return self.dec2(self.dec1(z.to(self.dtype))), z, log_variance, mean


def toy_loss_fn(model, video, mask, rngs, kl_weight=KL_WEIGHT, train=True):
    """mean_b MSE_b + kl_weight * mean_b KL_b -> (loss, aux)."""
    recon, _z, log_variance, mean = model(video, mask, rngs, train=train)
    b, t = mask.shape
    mse, _ = ops.masked_mse_mae(video, recon, mask, 1)
    kl = model._kl if model._kl is not None else ops.kl_per_sample(mean, log_variance, mask.reshape(b, t))
    loss = mse.mean() + kl_weight * kl.mean()
    return loss, {"MSE": mse.mean().detach(), "kl_loss": kl.mean().detach(), "reconstruction": recon.detach()}


def toy_train_step(model, optimizer, video, mask, rngs, kl_weight=KL_WEIGHT):
    """value_and_grad + optimizer.update, as the reference's train_step (train/rl_nonadversarial.py:188-198)."""
    optimizer.zero_grad()
    loss, aux = toy_loss_fn(model, video, mask, rngs, kl_weight)
    loss.backward()
    optimizer.update()
    return loss.detach(), aux
