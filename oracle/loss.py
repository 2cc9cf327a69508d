"""CPU restatement of the recon+KL losses.  Test infrastructure only.

``loss_fn_rl``   follows train/rl_nonadversarial.py:100-186 (pair-doubled, REINFORCE term).
``loss_fn_plain`` follows train/legacy/training_loop_adversarial.py:90-124 (MSE + sel + KL).
The perceptual (VGG16) term needs remote weights and is out of scope
(SURVEY.md section 2.1); callers pass ``perceptual=None`` => zeros, as the
reference's own CPU test does (claude_distributed/test_training_loop.py:71).
"""
import torch
from einops import rearrange, reduce

HPARAMS = {  # train/rl_nonadversarial.py:46-57,255-263
    "gamma1": 0.2, "gamma2": 0.001, "gamma3": 0.1, "gamma4": 0.05,
    "max_compression_rate": 2, "magnify_negatives_rate": 100, "rl_loss_weight": 0.01,
}


def per_sample_mean(x):
    """rl_nonadversarial.py:59-60."""
    return x.mean(dim=tuple(range(1, x.ndim)))


def magnify_negatives(x, rate):
    """rl_nonadversarial.py:69-71."""
    return torch.where(x < 0, x * rate, x)


def expand_mask(mask_bt, hw):
    """train_step's mask expansion (b,t) -> (b*hw,1,1,t).  rl_nonadversarial.py:190-192."""
    b, t = mask_bt.shape
    return mask_bt[:, None, :].expand(b, hw, t).reshape(b * hw, 1, 1, t)


def _q(x, dtype):
    return x if dtype is None or dtype == torch.float32 else x.to(dtype).to(torch.float32)


def masked_mse_mae(video, recon, mask_bt, dtype=torch.float32):
    """Per-sample masked MSE and MAE.  rl_nonadversarial.py:104-121.

    mean over (h,w,c) of [ sum_t ((v-r)*m_t)^2 / len ]; same with abs.  ``dtype``: the compute dtype the driver casts the clip to
    (rl_nonadversarial.py:330) and the model returns: ``video - reconstruction`` is an array of that dtype (rounded), the float32
    mask promotes what follows.
    """
    lens = torch.clamp(mask_bt.sum(dim=1, keepdim=True), min=1.0)
    m = mask_bt[:, :, None, None, None]
    ln = lens[:, :, None, None, None]
    err = _q(video - recon, dtype) * m
    mae = per_sample_mean(err.abs().sum(dim=1, keepdim=True) / ln)
    mse = per_sample_mean((err * err).sum(dim=1, keepdim=True) / ln)
    return mse, mae


def kl_per_sample(mean, logvar, mask_bt, dtype=torch.float32):
    """0.5*(exp(lv) - 1 - lv + mu^2) * m_t / len, mean over (t,hw,c).  rl_nonadversarial.py:144-147.

    ``dtype``: the model's compute dtype.  ``logvar`` and ``mean`` are arrays of that dtype, so every elementwise step up to the
    ``0.5 *`` is one (each result rounded); the float32 mask then promotes the product (rl_nonadversarial.py:146)."""
    lens = torch.clamp(mask_bt.sum(dim=1, keepdim=True), min=1.0)
    m = mask_bt[:, :, None, None]
    inner = _q(_q(_q(_q(torch.exp(logvar), dtype) - 1, dtype) - logvar, dtype) + _q(mean * mean, dtype), dtype)
    kl = _q(0.5 * inner, dtype) * m / lens[:, :, None, None]
    return per_sample_mean(kl)


def loss_fn_rl(outputs, video, original_mask, hparams=HPARAMS, perceptual=None, dtype=torch.float32):
    """rl_nonadversarial.py:100-186.  ``outputs`` is the 6-tuple of rl_model.VideoVAE."""
    recon, _comp, selection, selection_mask, logvar, mean = outputs
    om = original_mask.to(torch.float32).repeat_interleave(2, dim=0)      # :104
    lens = torch.clamp(om.sum(dim=1, keepdim=True), min=1.0)              # :105-106
    video2 = video.repeat_interleave(2, dim=0)                            # :110
    mse, mae = masked_mse_mae(video2, recon, om, dtype)                   # :114-121
    perc = torch.zeros_like(mse) if perceptual is None else perceptual    # :125
    klm = om[:, :, None, None]                                            # :127
    selection_sum = reduce(selection_mask * klm, "b t 1 1 -> b 1", "sum")  # :130
    density = selection_sum / lens                                        # :133
    diff = density - (1 / hparams["max_compression_rate"])                # :139
    sel_loss = per_sample_mean(magnify_negatives(diff, hparams["magnify_negatives_rate"]) ** 2)  # :141
    kl = kl_per_sample(mean, logvar, om, dtype)                           # :146-147
    per_sample = (mse + hparams["gamma3"] * perc + hparams["gamma1"] * sel_loss
                  + hparams["gamma2"] * kl + hparams["gamma4"] * mae)     # :149
    pairs = rearrange(per_sample, "(b p) -> b p", p=2)                    # :150
    means = pairs.mean(dim=1, keepdim=True)
    stds = pairs.std(dim=1, unbiased=False, keepdim=True) + 1e-6          # jnp.std is population std, :152
    dis = (pairs - means) / stds                                          # :153
    actions = rearrange(selection_mask, "(b p) t 1 1 -> b p t", p=2)      # :154
    sel = rearrange(selection, "(b p) t 1 1 -> b p t", p=2)               # :157
    raw = torch.clamp((sel + actions - 1).abs(), 1e-6, 1.0 - 1e-6)        # :163
    probs = raw / raw.detach()                                            # :164
    rl_mask = rearrange(om, "(b p) t -> b p t", p=2).to(torch.bool)       # :165
    probs = torch.where(rl_mask, probs, torch.ones(()))                   # :166
    raw_m = torch.where(rl_mask, raw, torch.ones(()))                     # :168
    traj = raw_m.prod(dim=2, keepdim=True)                                # :169
    probs = probs.prod(dim=2, keepdim=True)                               # :171
    rl_loss = probs * dis.detach()[:, :, None]                            # :172-173
    loss = per_sample.mean() + rl_loss.mean() * hparams["rl_loss_weight"]  # :174
    aux = {
        "MSE": mse.mean(), "perceptual_loss": perc.mean(), "selection_loss": sel_loss.mean(),
        "kl_loss": kl.mean(), "reconstruction": recon, "kept_frame_density": density.mean(),
        "mean_trajectory_prob": traj.mean(), "rl_loss": rl_loss.mean(), "per_sample_MAE": mae.mean(),
    }
    return loss, aux


def loss_fn_plain(outputs, video, original_mask, hparams=HPARAMS, dtype=torch.float32):
    """legacy/training_loop_adversarial.py:90-124.  ``outputs`` is the 5-tuple of model.VideoVAE."""
    recon, _comp, selection, logvar, mean = outputs
    om = original_mask.to(torch.float32)
    lens = torch.clamp(om.sum(dim=1, keepdim=True), min=1.0)              # :94-95
    mse_ps, _ = masked_mse_mae(video, recon, om, dtype)                   # :97-101
    mse = mse_ps.mean()                                                   # :102 (equal-size samples)
    klm = om[:, :, None, None]                                            # :104
    selection_sum = reduce(selection * klm, "b t 1 1 -> b 1", "sum")      # :106
    density = selection_sum / lens                                        # :109
    diff = density - (1 / hparams["max_compression_rate"])                # :115
    sel_loss = (magnify_negatives(diff, hparams["magnify_negatives_rate"]) ** 2).mean()  # :117
    kl = kl_per_sample(mean, logvar, om, dtype).mean()                    # :119-121
    loss = mse + hparams["gamma1"] * sel_loss + hparams["gamma2"] * kl    # :122
    aux = {"MSE": mse, "selection_loss": sel_loss, "kl_loss": kl, "reconstruction": recon,
           "kept_frame_density": density.mean()}
    return loss, aux
