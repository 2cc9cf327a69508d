"""Encoder / Decoder / VideoVAE with the surface of the reference's train/rl_model.py (6-tuple, Bernoulli pairs)."""
import torch
from einops import rearrange
from torch import nn

from . import ops
from .layers import linear_pair
from .model import Encoder as _Encoder, Decoder, frame_mask  # Decoder is identical in both flavours (rl_model.py:62-97)

__all__ = ["Encoder", "Decoder", "VideoVAE"]

FUSED_HEADS = [True]      # test switch: False = the unfused heads (framework ops + ops.reparameterise_kl + ops.rl_gate)


class Encoder(_Encoder):
    """Reference train/rl_model.py:15-60 -> (mean, log_variance, selection probability (b,t,1))."""

    flavour = "rl"

    def forward(self, x, mask, rngs, train=True):
        mean, log_variance, logits = self._trunk(x, mask)
        return mean, log_variance, torch.sigmoid(logits)


class VideoVAE(nn.Module):
    """Reference train/rl_model.py:101-147 -> (reconstruction, compressed_representation, selection, selection_mask,
    log_variance, mean), every output pair-doubled along batch (samples 2k, 2k+1 share an input clip)."""

    def __init__(self, height, width, channels, patch_size, encoder_depth, decoder_depth, mlp_dim, num_heads, qkv_features,
                 max_temporal_len, spatial_compression_rate, unembedding_upsample_rate, rngs, dtype=torch.bfloat16,
                 param_dtype=torch.float32):
        super().__init__()
        key = rngs.sampling()
        self.encoder = Encoder(height, width, channels, patch_size, encoder_depth, mlp_dim, num_heads, qkv_features,
                               max_temporal_len, spatial_compression_rate, rngs, dtype, param_dtype)
        self.decoder = Decoder(height, width, channels, patch_size, decoder_depth, mlp_dim, num_heads, qkv_features,
                               max_temporal_len, spatial_compression_rate, unembedding_upsample_rate, rngs, dtype, param_dtype)
        ld = channels * patch_size * patch_size // spatial_compression_rate
        self.fill_token = nn.Parameter(torch.randn((1, 1, 1, ld), generator=key.generator("cpu")) * 0.02)

    def forward(self, x, mask, rngs, train=True):
        self._kl = None
        enc = self.encoder
        if train and FUSED_HEADS[0] and type(enc) is Encoder and enc.gated_ok(x, self.fill_token):
            # GPU train step: everything behind the encoder's two 768 -> ld products -- softplus / log, both selection layers, the sigmoid, the
            # reparameterisation, the KL term, the pair doubling of mean / log-variance / selection, the Bernoulli frame masks and the latent gate --
            # in one launch each way (ops.encoder_head_rl).  The noise draws keep the reference's order: reparameterisation, then Bernoulli.
            sl1, sl2 = enc.selection_layer1, enc.selection_layer2
            h = enc._features(x, mask)
            mean1, v = linear_pair(h, enc.spatial_compression, enc.variance_estimator)
            b, t = mean1.shape[0], mean1.shape[1]
            eps = rngs.draw("reparam_eps", "normal", mean1.shape, mean1.device)
            u = rngs.draw("bernoulli_u", "uniform", (2 * b, t, 1, 1), mean1.device)
            log_variance, mean, compressed_representation, selection, selection_mask, kl2 = ops.encoder_head_rl(
                mean1, v, sl1.kernel, sl1.bias, sl2.kernel, sl2.bias, self.fill_token, u, eps, frame_mask(mask, b, t))
            self._kl = (mean, log_variance, kl2)          # (2b, t) per-frame partial sums: the rl loss tail adds a sample's up itself
            # the decoder gets the UN-doubled mask: its temporal attention broadcasts a mask row over the consecutive sequences that share it
            # (layers.Attention: div = sequences // mask rows), and the members of a pair are consecutive samples
            reconstruction = self.decoder(compressed_representation, mask, rngs, train=train)
            return reconstruction, compressed_representation, selection, selection_mask, log_variance, mean
        mean, log_variance, selection = self.encoder(x, mask, rngs, train=train)
        kl = None
        if train:
            noise = rngs.draw("reparam_eps", "normal", log_variance.shape, log_variance.device)
            sampled_latent, kl = ops.reparameterise_kl(mean, log_variance, noise, frame_mask(mask, mean.shape[0], mean.shape[1]))
        else:
            sampled_latent = mean
        prob = selection
        selection = rearrange(selection, "b t 1 -> b t 1 1").repeat_interleave(2, dim=0)
        fused = (self.decoder.dtype == torch.bfloat16 and ops.rl_gate_ok(sampled_latent, prob, self.fill_token))
        if not fused:
            sampled_latent = sampled_latent.repeat_interleave(2, dim=0)
        mean = mean.repeat_interleave(2, dim=0)
        log_variance = log_variance.repeat_interleave(2, dim=0)
        if kl is not None:                                  # both members of a pair share mean / log-variance / mask: same KL term
            self._kl = (mean, log_variance, kl.repeat_interleave(2, dim=0))
        mask = mask.repeat_interleave(2, dim=0)
        u = rngs.draw("bernoulli_u", "uniform", selection.shape, selection.device)
        if fused:
            # GPU train step: doubling, the Bernoulli masks and the gate in one launch; comp holds the decoder's compute dtype (the
            # reference's fp32 sum rounded once, which is what its decoder's first Linear does to it)
            compressed_representation, selection_mask = ops.rl_gate(sampled_latent, prob, u, self.fill_token)
        else:
            selection_mask = (u < selection).to(sampled_latent.dtype)
            compressed_representation = self.fill_token * (1 - selection_mask) + sampled_latent * selection_mask
        reconstruction = self.decoder(compressed_representation, mask, rngs, train=train)
        return reconstruction, compressed_representation, selection, selection_mask, log_variance, mean
