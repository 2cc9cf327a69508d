"""video_vae_amd: MI355X-native training path for the video VAE of floatingtrees/video-VAE.

Host code is Python on PyTorch-ROCm; the hot operators are hand-written HIP kernels (csrc/, built into
libvvae_hip.so, C ABI in include/vvae_hip.h).  Same class surface as the reference's train/model.py,
train/rl_model.py, train/unet.py, train/layers.py, train/model_loader.py.
"""
from .rngs import Rngs  # noqa: F401
from .unet import UNet, ConvBlock3D, DownBlock3D, UpBlock3D  # noqa: F401
from .layers import (PatchEmbedding, PatchUnEmbedding, RotaryEmbedding, Attention, MLP, FactoredAttention,  # noqa: F401
                     round_ste, GumbelSigmoidSTE)
from .model import Encoder, Decoder, VideoVAE  # noqa: F401
from .model_loader import load_checkpoint, save_checkpoint  # noqa: F401
from . import rl_model, loss, optim, ddp, ops, perceptual, classifier  # noqa: F401
