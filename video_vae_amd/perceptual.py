"""VGG16 perceptual loss on relu1_1 / relu1_2 / relu2_1 with the surface of the reference's train/vgg_tests.py.

    model, params = load_vgg(pretrained=None)                   # vgg_tests.py:8-32 (flaxmodels.VGG16(output='activations', include_head=False))
    ploss = get_adversarial_perceptual_loss_fn(model)           # vgg_tests.py:37-68: (params, x, target) -> (b,) per-sample loss
    loss, aux = loss_fn(model, video, mask, original_mask, rngs, hparams, ploss, params)      # rl_nonadversarial.py:125

Only the three convolutions the loss reads are built (conv1_1 3->64, conv1_2 64->64, 2x2 max-pool, conv2_1 64->128, each 3x3 SAME
+ bias + ReLU; ImageNet mean / std normalisation of [0, 1] inputs).  Frames are convolved where they lie: a 3x3 2-D convolution
over every frame of (b, t, h, w, c) IS an NDHWC Conv3d with a (1, 3, 3) kernel, so the layers run on the library's conv / pool
kernels and the squared feature differences on the masked-MSE kernel -- no "(b t) h w c" reshuffle.

Weights.  The reference downloads ImageNet weights through flaxmodels (``pretrained='imagenet'``); there is no network here, so
``pretrained=None`` (random init, an option of the reference's own loader) is the default, and ``pretrained=<path>`` loads a
``.npz`` / ``.pt`` holding ``conv1_1.weight`` (3, 3, 3, 64) HWIO, ``conv1_1.bias`` ... ``conv2_1.bias`` for users who have them.
"""
import numpy as np
import torch
from torch import nn

from . import ops
from .rngs import Rngs, truncated_normal_

PERCEPTUAL_LAYERS = ("relu1_1", "relu1_2", "relu2_1")
_SHAPES = {"conv1_1": (3, 64), "conv1_2": (64, 64), "conv2_1": (64, 128)}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class VGG16Features(nn.Module):
    """The first three convolutions of VGG16; ``forward(params, x)`` -> {"relu1_1", "relu1_2", "relu2_1"} on (b, t, h, w, 3) in [0, 1]."""

    def __init__(self, normalize=True, dtype=torch.bfloat16):
        super().__init__()
        self.normalize, self.dtype = normalize, dtype

    def init(self, rngs=None):
        rngs = rngs if rngs is not None else Rngs(0)
        p = {}
        for name, (cin, cout) in _SHAPES.items():
            p[f"{name}.weight"] = truncated_normal_((3, 3, cin, cout), 9 * cin, rngs.params())
            p[f"{name}.bias"] = torch.zeros(cout)
        return p

    def _conv_relu(self, params, name, x):
        w = params[f"{name}.weight"]
        y = ops.conv3d(x, w.reshape(1, *w.shape), params[f"{name}.bias"])          # (1, 3, 3, Cin, Cout): per-frame 3x3 SAME
        return torch.relu(y)

    def forward(self, params, x):
        x = x.to(self.dtype)
        if self.normalize:
            mean = torch.tensor(IMAGENET_MEAN, device=x.device, dtype=self.dtype)
            std = torch.tensor(IMAGENET_STD, device=x.device, dtype=self.dtype)
            x = (x - mean) / std
        r11 = self._conv_relu(params, "conv1_1", x)
        r12 = self._conv_relu(params, "conv1_2", r11)
        r21 = self._conv_relu(params, "conv2_1", ops.max_pool_1x2x2(r12))
        return {"relu1_1": r11, "relu1_2": r12, "relu2_1": r21}

    apply = forward                                                  # flax spelling: model.apply(params, x)


def load_vgg(pretrained=None, normalize=True, device=None, dtype=torch.bfloat16):
    """-> (model, params).  ``pretrained``: None (random init) or a path to an .npz / .pt with the six tensors named above.
    Parameters are cast to the compute dtype as the reference does (vgg_tests.py:30)."""
    model = VGG16Features(normalize=normalize, dtype=dtype)
    if pretrained is None:
        params = model.init(Rngs(0))
    elif pretrained == "imagenet":
        raise ValueError("the ImageNet weights are a remote download in the reference (flaxmodels); pass a local file instead")
    elif str(pretrained).endswith(".npz"):
        with np.load(pretrained) as z:
            params = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    else:
        params = torch.load(pretrained, map_location="cpu", weights_only=True)
    want = {f"{n}.{s}" for n in _SHAPES for s in ("weight", "bias")}
    if set(params) != want:
        raise KeyError(f"VGG parameter names: expected {sorted(want)}, got {sorted(params)}")
    # kernels stay fp32 containers holding values rounded to the compute dtype (the conv kernels take fp32 Flax-layout weights)
    params = {k: v.to(dtype).to(torch.float32) for k, v in params.items()}
    if device is not None:
        params = {k: v.to(device) for k, v in params.items()}
    return model, params


def _feature_mse(fx, ft, target_div):
    """mean over (t, h, w, c) of (fx - ft)^2 per sample: the masked-MSE kernel with an all-ones mask (gradient flows to fx only).
    ``target_div``: fx sample i is compared with ft sample i // target_div."""
    ones = torch.ones(fx.shape[:2], dtype=torch.float32, device=fx.device)
    mse, _ = ops.masked_mse_mae(ft, fx, ones, target_div)
    return mse


def get_adversarial_perceptual_loss_fn(model):
    """(params, x, target) -> (b,): per frame, the sum over the three layers of the mean squared feature difference; then the mean
    over frames (vgg_tests.py:45-66).  Frames have equal sizes, so that is each layer's mean over (t, h, w, c) summed."""
    def perceptual_loss(params, x, target, target_div=1):
        """``target_div`` = 2 lets the rl driver pass the b clips once for its 2b pair-doubled reconstructions (the reference
        repeats the video first, rl_nonadversarial.py:110; the features of a repeated clip are the same features)."""
        fx = model(params, x)
        with torch.no_grad():
            ft = model(params, target)
        return sum(_feature_mse(fx[k], ft[k], target_div) for k in PERCEPTUAL_LAYERS)
    perceptual_loss.takes_target_div = True
    return perceptual_loss


def get_perceptual_loss_fn(model):
    """(params, x, target) -> scalar: the sum over the three layers of the mean squared feature difference (vgg_tests.py:70-97)."""
    per_sample = get_adversarial_perceptual_loss_fn(model)

    def perceptual_loss(params, x, target):
        return per_sample(params, x, target).mean()
    return perceptual_loss


def get_perceptual_loss(model, params, x, target):
    """vgg_tests.py:100-131."""
    return get_perceptual_loss_fn(model)(params, x, target)
