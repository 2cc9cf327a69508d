"""Encoder / Decoder / VideoVAE with the surface of the reference's train/model.py (5-tuple, Gumbel-STE gate)."""
import torch
import torch.nn.functional as F
from einops import rearrange
from torch import nn

from . import ops
from .layers import PatchEmbedding, FactoredAttention, GumbelSigmoidSTE, PatchUnEmbedding, Linear, linear_pair
from .rngs import Rngs
from .unet import UNet


def frame_mask(mask, b, t):
    """(b, t) float view of the temporal mask the drivers pass: (b*hw,1,1,t) as train_step expands it (rl_nonadversarial.py:190-192)
    or (b,1,1,t) (claude_distributed/layers.py:213-214); every patch of a sample carries the same row."""
    return mask.reshape(b, -1, t)[:, 0].to(torch.float32)


class Encoder(nn.Module):
    """Reference train/model.py:14-60 -> (mean, log_variance, selection (b,t,1,1) in {0,1})."""

    flavour = "model"

    def __init__(self, height, width, channels, patch_size, depth, mlp_dim, num_heads, qkv_features, max_temporal_len,
                 spatial_compression_rate, rngs, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        max_spatial_len = height // patch_size * width // patch_size
        self.last_dim = channels * patch_size * patch_size
        self.dtype = dtype
        self.patch_embedding = PatchEmbedding(height, width, channels, patch_size, rngs, dtype, param_dtype)
        ld = self.last_dim // spatial_compression_rate
        self.spatial_compression = Linear(self.last_dim, ld, rngs, dtype, param_dtype)
        self.variance_estimator = Linear(self.last_dim, ld, rngs, dtype, param_dtype)
        self.selection_layer1 = Linear(ld, 1, rngs, dtype, param_dtype)
        self.selection_layer2 = Linear(max_spatial_len, 1, rngs, dtype, param_dtype)
        self.gumbel_sigmoid = GumbelSigmoidSTE(temperature=1.0)
        self.layers = nn.ModuleList([
            FactoredAttention(mlp_dim, self.last_dim, num_heads, qkv_features, max_temporal_len, max_spatial_len, rngs,
                              dtype, param_dtype) for _ in range(depth)])

    def _features(self, x, mask):
        x = self.patch_embedding(x)
        # each layer hands its last residual add to the next layer's first LayerNorm kernel (layers.FactoredAttention)
        pend, n = None, len(self.layers)
        for i, layer in enumerate(self.layers):
            r = layer(x, mask, pending=pend, defer=i + 1 < n)
            if i + 1 < n:
                pend = r
            else:
                x = r
        return x

    def _trunk(self, x, mask):
        x = self._features(x, mask)
        mean, v = linear_pair(x, self.spatial_compression, self.variance_estimator)      # GPU: one node, the input gradients meet inside a product
        variance = F.softplus(v)
        log_variance = torch.log(variance)
        si = rearrange(self.selection_layer1(mean), "b t hw 1 -> b t hw")
        return mean, log_variance, self.selection_layer2(si) + 1

    def forward(self, x, mask, rngs, train=True):
        mean, log_variance, logits = self._trunk(x, mask)
        selection = self.gumbel_sigmoid(logits.float(), rngs, train=train)
        return mean, log_variance, rearrange(selection, "b t 1 -> b t 1 1")

    def gated_ok(self, x, fill_token):
        """May forward_gated run?  (bf16 GPU model, temperature 1, fp32 parameters, shapes the fused kernel covers.)"""
        sl1, sl2 = self.selection_layer1, self.selection_layer2
        b, t = x.shape[0], x.shape[1]
        hw, ld = sl2.kernel.shape[0], sl1.kernel.shape[0]
        return (self.dtype == torch.bfloat16 and x.is_cuda and self.gumbel_sigmoid.temperature == 1.0 and fill_token.numel() == ld
                and all(p.dtype == torch.float32 for p in (sl1.kernel, sl1.bias, sl2.kernel, sl2.bias, fill_token))
                and bool(ops.lib().vvae_encoder_head_ok(b, t, hw, ld)))

    def forward_gated(self, x, mask, rngs, fill_token):
        """Train mode on the GPU -> (mean, log_variance, selection, compressed_representation, kl (b, t)): everything behind the two
        768 -> ld products -- softplus / log, both selection layers, the Gumbel-sigmoid STE, the reparameterisation, the KL term and the
        latent gate fill * (1 - s) + z * s of VideoVAE.__call__ (model.py:121-133) -- in one launch each way (ops.encoder_head) instead of
        ~45 framework launches per step.  The noise draws keep the reference's order: Gumbel, then reparameterisation."""
        sl1, sl2 = self.selection_layer1, self.selection_layer2
        h = self._features(x, mask)
        mean, v = linear_pair(h, self.spatial_compression, self.variance_estimator)      # one node: the two input gradients meet inside a product
        b, t = mean.shape[0], mean.shape[1]
        u = rngs.draw("gumbel_u", "uniform", (b, t, 1), mean.device)
        eps = rngs.draw("reparam_eps", "normal", mean.shape, mean.device)
        log_variance, comp, selection, kl = ops.encoder_head(mean, v, sl1.kernel, sl1.bias, sl2.kernel, sl2.bias, fill_token, u, eps,
                                                             frame_mask(mask, b, t))
        return mean, log_variance, selection, comp, kl


class Decoder(nn.Module):
    """Reference train/model.py:62-97: Linear -> depth x FactoredAttention -> un-patchify -> coarse + UNet(features)."""

    def __init__(self, height, width, channels, patch_size, depth, mlp_dim, num_heads, qkv_features, max_temporal_len,
                 spatial_compression_rate, unembedding_upsample_rate, rngs, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.last_dim = channels * patch_size * patch_size
        self.dtype = dtype
        self.patch_unembedding = PatchUnEmbedding(height, width, channels, patch_size, unembedding_upsample_rate, rngs,
                                                  dtype, param_dtype)
        self.spatial_decompression = Linear(self.last_dim // spatial_compression_rate, self.last_dim, rngs, dtype, param_dtype)
        max_spatial_len = height // patch_size * width // patch_size
        self.layers = nn.ModuleList([
            FactoredAttention(mlp_dim, self.last_dim, num_heads, qkv_features, max_temporal_len, max_spatial_len, rngs,
                              dtype, param_dtype) for _ in range(depth)])
        self.unet = UNet(channels=channels * unembedding_upsample_rate, base_features=16, num_levels=3,
                         out_features=channels, rngs=rngs, dtype=dtype, param_dtype=param_dtype)

    def forward(self, x, mask, rngs, train=True):
        x = self.spatial_decompression(x)
        # each layer hands its last residual add to the next layer's first LayerNorm kernel (layers.FactoredAttention)
        pend, n = None, len(self.layers)
        for i, layer in enumerate(self.layers):
            r = layer(x, mask, pending=pend, defer=i + 1 < n)
            if i + 1 < n:
                pend = r
            else:
                x = r
        plan = self.unet.pad_plan(x.is_cuda)                        # the UNet's 12-channel weights ride in the un-embedding's pad launch
        out = self.patch_unembedding.forward_padded(x, plan)       # bf16 GPU path: un-patchify + channel pad in one copy
        if out is not None and plan is not None:
            feat, x, padded = out
            return self.unet(feat, residual=x, padded=padded)
        feat, x = out if out is not None else self.patch_unembedding(x)
        return self.unet(feat, residual=x)                          # x + self.unet(feat): the add rides in the UNet's final product


class VideoVAE(nn.Module):
    """Reference train/model.py:101-136 -> (reconstruction, compressed_representation, selection, log_variance, mean)."""

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
        if train and type(self.encoder) is Encoder and self.encoder.gated_ok(x, self.fill_token):
            # GPU train step: heads, noise, KL and gate come out of one kernel; compressed_representation holds the decoder's compute dtype
            # (the reference's fp32 sum rounded once, which is what its decoder's first Linear does to it: the same numbers downstream)
            mean, log_variance, selection, compressed_representation, kl = self.encoder.forward_gated(x, mask, rngs, self.fill_token)
            self._kl = (mean, log_variance, kl)
            reconstruction = self.decoder(compressed_representation, mask, rngs, train=train)
            return reconstruction, compressed_representation, selection, log_variance, mean
        mean, log_variance, selection = self.encoder(x, mask, rngs, train=train)
        if train:
            noise = rngs.draw("reparam_eps", "normal", log_variance.shape, log_variance.device)
            # z and the per-sample KL term in ONE pass over (mean, log_variance): the loss picks the KL up from here
            # (loss.kl_from_model) instead of streaming the two tensors a second time
            sampled_latent, kl = ops.reparameterise_kl(mean, log_variance, noise, frame_mask(mask, mean.shape[0], mean.shape[1]))
            self._kl = (mean, log_variance, kl)
        else:
            sampled_latent = mean
        # fill * (1 - s) + z * s as the reference writes it (model.py:133).  torch.lerp(fill, z, s) would be one launch instead of four,
        # but its backward adds a multi-block framework reduction (the gradient of the broadcast fill_token) to the captured step, and
        # those are kept out of a replayed hipGraph (the reduction's semaphore memset is a graph memset node).
        compressed_representation = self.fill_token * (1 - selection) + sampled_latent * selection
        reconstruction = self.decoder(compressed_representation, mask, rngs, train=train)
        return reconstruction, compressed_representation, selection, log_variance, mean
