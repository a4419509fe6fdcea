"""Minnen'20 hyper-encoder / hyper-decoder on HIP kernels.

Mirrors src/models/subnet/hyperprior/minnen20_hyperprior.py:8-55 (module tree + arithmetic).
ReLU is fused into the producing (transposed) convolution; the two decoder branches write straight
into the halves of one [N, hyper_out_ch, H, W] buffer instead of a torch.cat.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .layers import Conv2d, ConvTranspose2d
from .registry import HYPERDECODER_REGISTRY, HYPERENCODER_REGISTRY

Tensor = torch.Tensor


@HYPERENCODER_REGISTRY.register()
class Minnen20HyperEncoder(nn.Module):
    def __init__(self, bottleneck_y: int = 320, bottleneck_z: int = 192):
        super().__init__()
        self.n_downsampling_layers = 2
        self.conv1 = Conv2d(bottleneck_y, 320, 3, 1, 1)
        self.conv2 = Conv2d(320, 256, 5, 2, 2)
        self.conv3 = Conv2d(256, bottleneck_z, 5, 2, 2)

    def forward(self, x: Tensor) -> Tensor:
        x = self.conv1(x, act=ops.ACT_RELU)
        x = self.conv2(x, act=ops.ACT_RELU)
        return self.conv3(x)


class HyperDecoderBlock(nn.Module):
    def __init__(self, in_ch: int = 192, out_ch: int = 320):
        super().__init__()
        self.conv1 = ConvTranspose2d(in_ch, 192, 5, 2, 2, 1)
        self.conv2 = ConvTranspose2d(192, 256, 5, 2, 2, 1)
        self.conv3 = ConvTranspose2d(256, out_ch, 3, 1, 1)

    def forward(self, x: Tensor, out=None) -> Tensor:
        x = self.conv1(x, act=ops.ACT_RELU)
        x = self.conv2(x, act=ops.ACT_RELU)
        return self.conv3(x, out=out)


@HYPERDECODER_REGISTRY.register()
class Minnen20HyperDecoder(nn.Module):
    def __init__(self, bottleneck_z: int = 192, hyper_out_ch: int = 640):
        super().__init__()
        assert hyper_out_ch % 2 == 0
        self.half = hyper_out_ch // 2
        self.hd_mu = HyperDecoderBlock(bottleneck_z, self.half)
        self.hd_std = HyperDecoderBlock(bottleneck_z, self.half)

    def forward(self, x: Tensor) -> Tensor:
        N, _, H, W = x.shape
        out = torch.empty((N, 2 * self.half, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
        self.hd_mu(x, out=out[:, : self.half])
        self.hd_std(x, out=out[:, self.half:])
        return out
