"""CPU restatement of the VGG16 perceptual loss (train/vgg_tests.py:8-131).  Test infrastructure only.

flaxmodels.VGG16(output='activations', include_head=False, normalize=True, dtype=bf16) on (n, h, w, 3) inputs in [0, 1]:
x = (x - mean) / std with the ImageNet statistics, then conv1_1 (3->64), conv1_2 (64->64), 2x2/2 max-pool, conv2_1 (64->128),
every conv 3x3 SAME + bias + ReLU.  flaxmodels is not installed (PARITY UNPINNED, see oracle/__init__.py): the layer recipe is the
published VGG16 one; kernels are HWIO as in Flax.
"""
import torch
import torch.nn.functional as F

from . import nn as O

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def conv3x3_relu(x, w, b, dtype):
    """(n, h, w, cin) -> (n, h, w, cout): cross-correlation, SAME, bias, ReLU; w (3, 3, cin, cout)."""
    y = F.conv2d(O.q(x, dtype).permute(0, 3, 1, 2), O.q(w, dtype).permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    y = O.q(O.q(y, dtype) + O.q(b, dtype), dtype)
    return torch.relu(y)


def vgg_features(p, x, dtype=O.F32, normalize=True):
    x = O.q(x, dtype)
    if normalize:
        x = O.q(O.q(x - O.q(torch.tensor(MEAN), dtype), dtype) / O.q(torch.tensor(STD), dtype), dtype)
    r11 = conv3x3_relu(x, p["conv1_1.weight"], p["conv1_1.bias"], dtype)
    r12 = conv3x3_relu(r11, p["conv1_2.weight"], p["conv1_2.bias"], dtype)
    pool = F.max_pool2d(r12.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    r21 = conv3x3_relu(pool, p["conv2_1.weight"], p["conv2_1.bias"], dtype)
    return {"relu1_1": r11, "relu1_2": r12, "relu2_1": r21}


def adversarial_perceptual_loss(p, x, target, dtype=O.F32):
    """vgg_tests.py:45-66: (b, t, h, w, c) x 2 -> (b,)."""
    b, t = x.shape[:2]
    fx = vgg_features(p, x.reshape(b * t, *x.shape[2:]), dtype)
    ft = vgg_features(p, target.reshape(b * t, *target.shape[2:]), dtype)
    per_frame = sum(((fx[k] - ft[k]) ** 2).mean(dim=(1, 2, 3)) for k in ("relu1_1", "relu1_2", "relu2_1"))
    return per_frame.reshape(b, t).mean(dim=-1)


def perceptual_loss(p, x, target, dtype=O.F32):
    """vgg_tests.py:70-97: scalar."""
    b, t = x.shape[:2]
    fx = vgg_features(p, x.reshape(b * t, *x.shape[2:]), dtype)
    ft = vgg_features(p, target.reshape(b * t, *target.shape[2:]), dtype)
    return sum(((fx[k] - ft[k]) ** 2).mean() for k in ("relu1_1", "relu1_2", "relu2_1"))
