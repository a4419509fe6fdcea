"""Swin-transformer VQ-index estimator on HIP kernels.

Mirrors (module tree + arithmetic) src/models/subnet/vq_estimator/swin_vq_estimator.py:16-98,
src/models/layer/swinir_layers.py:17-33 (Mlp), :68-148 (WindowAttention), :167-282
(SwinTransformerBlock), :352-408 (BasicLayer), :422-485 (RSTB) and
src/models/layer/femasr_layers.py:65-84 (ResBlock with GroupNorm(32, eps 1e-6) + SiLU).
The token tensor [B, H*W, C] of the reference is kept as the NCHW map [B, C, H, W]: LayerNorm runs
over the channel axis per pixel, nn.Linear layers are 1x1 convolutions, window partition / cyclic
shift / mask are index arithmetic inside the window-attention kernel (csrc/swin.hip).  The shift
masks the reference rebuilds on the CPU every call are never materialised.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn as nn

from . import ops
from .layers import Act, Conv2d, GroupNorm, LayerNormC, Linear
from .registry import VQ_ESTIMATOR_REGISTRY

Tensor = torch.Tensor


class NormLayer(nn.Module):
    """femasr_layers.py:5-29 with norm_type 'gn'."""

    def __init__(self, channels: int, norm_type: str = "gn"):
        super().__init__()
        assert norm_type == "gn"
        self.norm = GroupNorm(channels, 32, 1e-6)


class ResBlock(nn.Module):
    """femasr_layers.py:65-84: (GN, SiLU, conv3x3) x 2 + input."""

    def __init__(self, in_channel: int, out_channel: int, norm_type: str = "gn", act_type: str = "silu"):
        super().__init__()
        assert act_type == "silu"
        self.conv = nn.Sequential(NormLayer(in_channel, norm_type), Act(), Conv2d(in_channel, out_channel, 3, 1, 1),
                                  NormLayer(out_channel, norm_type), Act(), Conv2d(out_channel, out_channel, 3, 1, 1))

    def forward(self, x: Tensor) -> Tensor:
        h = self.conv[0].norm(x, act=ops.ACT_SWISH)
        h = self.conv[2](h)
        h = self.conv[3].norm(h, act=ops.ACT_SWISH, out=h)
        return self.conv[5](h, res=x)


class Mlp(nn.Module):
    def __init__(self, in_features: int, hidden_features: int):
        super().__init__()
        self.fc1 = Linear(in_features, hidden_features)
        self.fc2 = Linear(hidden_features, in_features)

    def forward(self, x: Tensor, res: Tensor) -> Tensor:
        return self.fc2(self.fc1(x, act=ops.ACT_GELU), res=res)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, window_size: Tuple[int, int], num_heads: int):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads), requires_grad=False)
        coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))     # kept for state-dict parity
        self.qkv = Linear(dim, dim * 3)
        self.proj = Linear(dim, dim)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim: int, input_resolution, num_heads: int, window_size: int = 7, shift_size: int = 0, mlp_ratio: float = 4.0):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(self.input_resolution) <= self.window_size:       # swinir_layers.py:188-191
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        self.norm1 = LayerNormC(dim)
        self.attn = WindowAttention(dim, (self.window_size, self.window_size), num_heads)
        self.norm2 = LayerNormC(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        if self.shift_size > 0:
            # state-dict parity only (the kernel derives the mask from coordinates)
            self.register_buffer("attn_mask", self._mask(tuple(self.input_resolution)))
        else:
            self.attn_mask = None

    def _mask(self, x_size) -> Tensor:
        H, W = x_size
        ws, sh = self.window_size, self.shift_size
        img = torch.zeros((1, H, W, 1))
        sl = (slice(0, -ws), slice(-ws, -sh), slice(-sh, None))
        cnt = 0
        for h in sl:
            for w in sl:
                img[:, h, w, :] = cnt
                cnt += 1
        mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))

    def forward(self, x: Tensor) -> Tensor:
        """x: [B, C, H, W] (H, W multiples of the window)."""
        h = self.norm1(x)
        qkv = self.attn.qkv(h)
        a = ops.swin_attn(qkv, self.attn.relative_position_bias_table, self.num_heads, self.window_size, self.shift_size)
        x = self.attn.proj(a, res=x)                       # shortcut + attention
        return self.mlp(self.norm2(x), res=x)              # x + mlp(norm2(x))


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if (i % 2 == 0) else window_size // 2, mlp_ratio)
            for i in range(depth)])

    def forward(self, x: Tensor) -> Tensor:
        for blk in self.blocks:
            x = blk(x)
        return x


class RSTB(nn.Module):
    """swinir_layers.py:422-485 with resi_connection '1conv': conv3x3(residual_group(x)) + x."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, **kwargs):
        super().__init__()
        self.residual_group = BasicLayer(dim, input_resolution, depth, num_heads, window_size, mlp_ratio)
        self.conv = Conv2d(dim, dim, 3, 1, 1)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv(self.residual_group(x), res=x)


@VQ_ESTIMATOR_REGISTRY.register()
class DualBlockSwinVqEstimator(nn.Module):
    """swin_vq_estimator.py:16-98."""

    def __init__(self, input_resolution=(32, 32), in_ch: int = 192, main_ch: int = 256, n_embed: int = 256, embed_dim: int = 4,
                 blk_depth: int = 6, num_heads: int = 8, window_size: int = 8, num_swin_blocks: int = 4, act_type: str = "silu",
                 norm_type: str = "gn", use_upsample: bool = False, rstb_kwargs: dict = {}, proj_pos: str = "before_rstb"):
        super().__init__()
        assert not use_upsample and proj_pos == "before_rstb"
        self.window_size = window_size
        self.first_block = nn.Sequential(Conv2d(in_ch, main_ch, 3, 1, 1), nn.Identity(), ResBlock(main_ch, main_ch, norm_type, act_type),
                                         ResBlock(main_ch, main_ch, norm_type, act_type), Conv2d(main_ch, main_ch, 3, 1, 1))
        self.embed_projection = Conv2d(main_ch, embed_dim, 1)
        self.swin_blks = nn.ModuleList([RSTB(main_ch, list(input_resolution), blk_depth, num_heads, window_size, **rstb_kwargs)
                                        for _ in range(num_swin_blocks)])
        self.out_block = nn.Sequential(ResBlock(main_ch, main_ch, norm_type, act_type), Conv2d(main_ch, n_embed, 3, 1, 1))

    def forward(self, x: Tensor, want_embed: bool = False):
        x = self.first_block[0](x)
        x = self.first_block[2](x)
        x = self.first_block[3](x)
        x = self.first_block[4](x)
        pred_embed = self.embed_projection(x) if want_embed else None     # training-loss head; skipped at inference
        b, c, h, w = x.shape
        ws = self.window_size
        pad_h = math.ceil(h / ws) * ws - h
        pad_w = math.ceil(w / ws) * ws - w
        if pad_h or pad_w:
            x = ops.pad_reflect(x, pad_h, pad_w)          # F.pad(..., mode='reflect') swin_vq_estimator.py:81
        for m in self.swin_blks:
            x = m(x)
        if pad_h or pad_w:
            x = ops.crop(x, h, w)
        x = self.out_block[0](x)
        logits = self.out_block[1](x)
        return pred_embed, logits
