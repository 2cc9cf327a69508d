"""recon+KL losses and the train / eval step, counterparts of the reference's drivers.

``loss_fn``        : train/rl_nonadversarial.py:100-186 (pair-doubled rl_model outputs, REINFORCE term).
``loss_fn_plain``  : train/legacy/training_loop_adversarial.py:90-124 (model.py outputs: MSE + selection + KL).
The heavy reductions (masked MSE/MAE over the clip, KL over the latent) are fused HIP kernels; the per-sample
scalar algebra stays in torch.  The VGG16 perceptual term is video_vae_amd/perceptual.py (``get_adversarial_perceptual_loss_fn``);
``perceptual_loss_fn=None`` gives zeros, as the reference's CPU test does (claude_distributed/test_training_loop.py:71).
"""
import torch
from einops import rearrange, reduce, repeat

from . import ops

HPARAMS = {  # train/rl_nonadversarial.py:46-57,255-263
    "gamma1": 0.2, "gamma2": 0.001, "gamma3": 0.1, "gamma4": 0.05,
    "max_compression_rate": 2, "magnify_negatives_rate": 100, "rl_loss_weight": 0.01,
}


def per_sample_mean(x):
    return x.mean(dim=tuple(range(1, x.ndim)))


def magnify_negatives(x, magnification_rate):
    return torch.where(x < 0, x * magnification_rate, x)


def expand_mask(mask, hw):
    """(b, t) -> (b*hw, 1, 1, t), as train_step does (rl_nonadversarial.py:190-192)."""
    mask = rearrange(mask, "b time -> b 1 1 time")
    mask = repeat(mask, "b 1 1 time -> b hw 1 1 time", hw=hw)
    return rearrange(mask, "b hw 1 1 time -> (b hw) 1 1 time")


def compact_mask(mask):
    """(b, t) -> (b, 1, 1, t): the un-expanded form claude_distributed/layers.py:213-214 takes.  Every block of this package accepts both
    forms (layers.FactoredAttention); this one is a view -- no copy and no (b*hw, t) conversion inside the captured step."""
    return mask.reshape(mask.shape[0], 1, 1, mask.shape[1])


def kl_from_model(model, mean, logvar, mask_bt):
    """Per-sample KL term (rl_nonadversarial.py:146-147).  A train-mode forward already produced it in the same pass that
    reparameterised (ops.reparameterise_kl) and left it on the model, keyed by the very tensors it returned; anything else
    (eval mode, a foreign model) takes the stand-alone kernel.  The fused encoder heads (model.Encoder.forward_gated) leave it as
    (b, t) partial sums, one per frame, which the loss tail adds up."""
    cached = getattr(model, "_kl", None)
    if cached is not None and cached[0] is mean and cached[1] is logvar and cached[2].shape[0] == mask_bt.shape[0]:
        return cached[2]
    return ops.kl_per_sample(mean, logvar, mask_bt)


def rl_loss_tail_ops(per_sample_error, per_sample_MAE, perceptual_loss, kl_loss, selection, selection_mask, output_mask, hparams):
    """The scalar end of loss_fn as framework ops (reference train/rl_nonadversarial.py:130-186): the CPU / fallback path, and what
    ops.rl_loss_tail is tested against."""
    sequence_lengths = torch.clamp(reduce(output_mask, "b time -> b 1", "sum"), min=1.0)
    kl_and_selection_mask = rearrange(output_mask, "b time -> b time 1 1")
    selection_sum = reduce(selection_mask * kl_and_selection_mask, "b time 1 1 -> b 1", "sum")
    kept_frame_density = selection_sum / sequence_lengths
    density_compression_difference = kept_frame_density - (1 / hparams["max_compression_rate"])
    selection_loss = per_sample_mean(torch.square(
        magnify_negatives(density_compression_difference, hparams["magnify_negatives_rate"])))

    per_sample_loss = (per_sample_error + hparams["gamma3"] * perceptual_loss + hparams["gamma1"] * selection_loss
                       + hparams["gamma2"] * kl_loss + hparams["gamma4"] * per_sample_MAE)
    pairs = rearrange(per_sample_loss, "(b p) -> b p", p=2)
    means = rearrange(per_sample_mean(pairs), "b -> b 1")
    stds = rearrange(pairs.std(dim=1, unbiased=False) + 1e-6, "b -> b 1")
    disadvantages = (pairs - means) / stds
    actions = rearrange(selection_mask, "(b p) time 1 1 -> b p time", p=2)
    selection = rearrange(selection, "(b p) time 1 1 -> b p time", p=2).float()
    raw_probs = torch.clamp(torch.abs(selection + actions - 1), 1e-6, 1.0 - 1e-6)
    probs = raw_probs / raw_probs.detach()
    rl_mask = rearrange(output_mask, "(b p) time -> b p time", p=2) > 0
    one = torch.ones((), device=probs.device, dtype=probs.dtype)
    probs = torch.where(rl_mask, probs, one)
    raw_trajectory_probs = torch.where(rl_mask, raw_probs, one).detach().prod(dim=2, keepdim=True)      # logged only
    # prod over time of factors that are all exactly 1.0 (x / stop_grad(x)): value 1, gradient sum_t d p_t -- written as
    # 1 + sum(p - 1), which is the same number and the same gradient but needs no ProdBackward (its zero check reads a scalar
    # back to the host, which a hipGraph capture of the step cannot contain)
    probs = 1.0 + (probs - 1.0).sum(dim=2, keepdim=True)
    rl_loss = probs * rearrange(disadvantages, "b p -> b p 1").detach()
    loss = per_sample_loss.mean() + rl_loss.mean() * hparams["rl_loss_weight"]
    return loss, {
        "MSE": per_sample_error.mean(), "perceptual_loss": perceptual_loss.mean(), "selection_loss": selection_loss.mean(),
        "kl_loss": kl_loss.mean(), "kept_frame_density": kept_frame_density.mean(),
        "mean_trajectory_prob": raw_trajectory_probs.mean(), "rl_loss": rl_loss.mean(), "per_sample_MAE": per_sample_MAE.mean(),
    }


def loss_fn(model, video, mask, original_mask, rngs, hparams, perceptual_loss_fn=None, vgg_params=None, train=True):
    reconstruction, _comp, selection, selection_mask, logvar, mean = model(video, mask, rngs, train=train)
    output_mask = original_mask.to(torch.float32).repeat_interleave(2, dim=0)
    kl_loss = kl_from_model(model, mean, logvar, output_mask)
    if perceptual_loss_fn is None and reconstruction.is_cuda:
        # GPU: the scalar end of this loss (and its backward) as ONE launch instead of ~100 framework kernels on (2b,) / (b, 2, t) tensors; the
        # per-workgroup partial sums of the MSE / MAE go straight to it (ops.rl_loss_tail adds a sample's up itself)
        mse_p, mae_p = ops.masked_mse_mae(video, reconstruction, output_mask, video_div=2, partials=True)
        if ops.rl_loss_tail_ok(mse_p, mae_p, kl_loss, selection, selection_mask, output_mask):
            loss, (MSE, perc, sel_l, kl_m, dens, traj, rl_m, MAE) = ops.rl_loss_tail(mse_p, mae_p, None, kl_loss, selection, selection_mask, output_mask,
                                                                                     hparams)
            return loss, {"MSE": MSE, "perceptual_loss": perc, "selection_loss": sel_l, "kl_loss": kl_m, "reconstruction": reconstruction,
                          "kept_frame_density": dens, "mean_trajectory_prob": traj, "rl_loss": rl_m, "per_sample_MAE": MAE}
        per_sample_error, per_sample_MAE = mse_p.sum(1), mae_p.sum(1)
    else:
        per_sample_error, per_sample_MAE = ops.masked_mse_mae(video, reconstruction, output_mask, video_div=2)
    if kl_loss.dim() == 2:                               # per-frame partial sums (ops.encoder_head_rl)
        kl_loss = kl_loss.sum(1)
    if perceptual_loss_fn is None:
        perceptual_loss = torch.zeros_like(per_sample_error)
    elif getattr(perceptual_loss_fn, "takes_target_div", False):
        perceptual_loss = perceptual_loss_fn(vgg_params, reconstruction, video, target_div=2)      # features of each clip once
    else:
        perceptual_loss = perceptual_loss_fn(vgg_params, reconstruction, video.repeat_interleave(2, dim=0))
    loss, aux = rl_loss_tail_ops(per_sample_error, per_sample_MAE, perceptual_loss, kl_loss, selection, selection_mask, output_mask, hparams)
    aux["reconstruction"] = reconstruction
    return loss, aux


def loss_fn_plain(model, video, mask, original_mask, rngs, hparams, train=True):
    reconstruction, _comp, selection, logvar, mean = model(video, mask, rngs, train=train)
    om = original_mask.to(torch.float32)
    # on the GPU the per-workgroup partial sums go straight to the loss tail, which adds a sample's up itself (no fold launch)
    mse_ps, _ = ops.masked_mse_mae(video, reconstruction, om, video_div=1, partials=reconstruction.is_cuda)
    kl_ps = kl_from_model(model, mean, logvar, om)
    if ops.plain_loss_tail_ok(mse_ps, kl_ps, selection, om):
        # GPU: the per-sample algebra below (and its backward) as ONE launch instead of ~45 framework kernels of a few bytes each
        loss, (MSE, selection_loss, kl_loss, density) = ops.plain_loss_tail(mse_ps, kl_ps, selection, om, hparams)
        return loss, {"MSE": MSE, "selection_loss": selection_loss, "kl_loss": kl_loss, "reconstruction": reconstruction,
                      "kept_frame_density": density}
    if kl_ps.dim() == 2:                                 # per-frame partial sums (ops.encoder_head)
        kl_ps = kl_ps.sum(1)
    if mse_ps.dim() == 2:
        mse_ps = mse_ps.sum(1)
    MSE = mse_ps.mean()
    sequence_lengths = torch.clamp(reduce(om, "b time -> b 1", "sum"), min=1.0)
    kl_and_selection_mask = rearrange(om, "b time -> b time 1 1")
    selection_sum = reduce(selection * kl_and_selection_mask, "b time 1 1 -> b 1", "sum")
    kept_frame_density = selection_sum / sequence_lengths
    diff = kept_frame_density - (1 / hparams["max_compression_rate"])
    selection_loss = torch.square(magnify_negatives(diff, hparams["magnify_negatives_rate"])).mean()
    kl_loss = kl_ps.mean()
    loss = MSE + hparams["gamma1"] * selection_loss + hparams["gamma2"] * kl_loss
    return loss, {"MSE": MSE, "selection_loss": selection_loss, "kl_loss": kl_loss, "reconstruction": reconstruction,
                  "kept_frame_density": kept_frame_density.mean()}


def _is_rl(model):
    return getattr(model.encoder, "flavour", "model") == "rl"


def train_step(model, optimizer, video, mask, hparams, hw, rngs, perceptual_loss_fn=None, vgg_params=None):
    """Counterpart of train_step (rl_nonadversarial.py:188-198): fwd + bwd + optimizer.update.

    ``optimizer`` is a ``video_vae_amd.optim.Optimizer``; with a DDP-wrapped model the gradient all-reduce
    overlaps this backward (see ddp.py).
    """
    original_mask = mask                               # nothing below writes to it (the reference's arrays are immutable)
    emask = compact_mask(mask)                         # what expand_mask(mask, hw) carries (rl_nonadversarial.py:190-192), as a view
    optimizer.zero_grad()
    if _is_rl(model):
        loss, aux = loss_fn(model, video, emask, original_mask, rngs, hparams, perceptual_loss_fn, vgg_params)
    else:
        loss, aux = loss_fn_plain(model, video, emask, original_mask, rngs, hparams)
    with ops.deferred_wgrad(optimizer):
        loss.backward(gradient=ops.unit_grad(loss) if (loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0) else None)
    optimizer.update()
    # detached: a caller that keeps the aux of step i while step i + 1 is captured as a hipGraph must not keep step i's autograd graph
    # (and with it AccumulateGrad nodes pinned to this stream) alive -- graph.GraphedTrainStep captures on a stream of its own
    return loss.detach(), {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in aux.items()}


@torch.no_grad()
def eval_step(model, video, mask, hparams, hw, rngs, perceptual_loss_fn=None, vgg_params=None):
    """eval_step calls the loss with train=True on purpose (rl_nonadversarial.py:200-208)."""
    original_mask = mask
    emask = compact_mask(mask)
    if _is_rl(model):
        return loss_fn(model, video, emask, original_mask, rngs, hparams, perceptual_loss_fn, vgg_params, train=True)
    return loss_fn_plain(model, video, emask, original_mask, rngs, hparams, train=True)
