"""VQGAN (f=8, 256 codes x 4 dims) encoder / decoder / quantiser on HIP kernels.

Mirrors the module tree (state-dict keys) and arithmetic of the vendored latent-diffusion code the
reference uses: ldm/modules/diffusionmodules/model.py:33-202 (nonlinearity, Normalize, Upsample,
Downsample, ResnetBlock, AttnBlock), :368-459 (Encoder), :462-568 (Decoder);
ldm/models/autoencoder.py:14-61,264-282 (VQModel / VQModelInterface);
taming/modules/vqvae/quantize.py:213-329 (VectorQuantizer2).
Fusions: GroupNorm+swish in one kernel; q/k/v 1x1 convs as one launch; nearest-x2 upsample and the
asymmetric (0,1,0,1) pad folded into the conv loader; residual adds in the conv epilogue; attention
as two fp32-MFMA batched GEMMs around a column softmax.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .layers import Conv2d, GroupNorm, allow_winograd

Tensor = torch.Tensor


class ResnetBlock(nn.Module):
    """ldm model.py:82-141 (temb None, dropout 0, nin_shortcut when channels change)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = GroupNorm(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = GroupNorm(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x: Tensor, out: Optional[Tensor] = None, in_part=None, want_part: bool = False) -> Tensor:
        """`in_part`: GroupNorm partial statistics of `x` from the convolution that produced it (or None); `want_part`: the caller feeds
        this block's output straight into another GroupNorm -- `self.out_part` then holds its statistics when conv2 ran on the F(4x4)
        kernel (decoder side only: the encoder's convolutions never do, its GroupNorms keep their own fp64 pass and their bits)."""
        h = self.norm1(x, act=ops.ACT_SWISH, part=in_part)
        h = self.conv1(h, gn_stats=True)
        h = self.norm2(h, act=ops.ACT_SWISH, out=h, part=self.conv1.take_gn_part())
        skip = self.nin_shortcut(x) if self.in_channels != self.out_channels else x
        y = self.conv2(h, res=skip, out=out, gn_stats=want_part)
        self.out_part = self.conv2.take_gn_part()
        return y


class AttnBlock(nn.Module):
    """ldm model.py:150-202: single-head attention over all H*W positions, c = channels."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = GroupNorm(in_channels)
        self.q = Conv2d(in_channels, in_channels, 1)
        self.k = Conv2d(in_channels, in_channels, 1)
        self.v = Conv2d(in_channels, in_channels, 1)
        self.proj_out = Conv2d(in_channels, in_channels, 1)
        self._qkv_plan = None
        self._qkv_key = None

    def _qkv(self):
        key = tuple((p.data_ptr(), p._version) for m in (self.q, self.k, self.v) for p in m.parameters())
        if self._qkv_plan is None or self._qkv_key != key:
            w = torch.cat([self.q.weight, self.k.weight, self.v.weight], 0)
            b = torch.cat([self.q.bias, self.k.bias, self.v.bias], 0)
            self._qkv_plan = ops.ConvPlan(w, b, "conv")
            self._qkv_key = key
        return self._qkv_plan

    @staticmethod
    def _attn_unfused(qkv: Tensor, N: int, Cc: int, H: int, W: int) -> Tensor:
        """Materialised-score form (two batched GEMMs around a column softmax) for token counts the fused kernel does not
        take (HW not a multiple of 64 never happens on the x64-padded path; kept for direct users of the block)."""
        HW = H * W
        bs = 3 * Cc * HW
        q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
        # scores stored [key j][query i]:  St[j][i] = c^-0.5 * sum_c k[c][j] q[c][i]   (model.py:186-188)
        St = torch.empty((N, HW, HW), dtype=torch.float32, device=qkv.device)
        ops.bgemm(k, (bs, 1, HW), q, (bs, HW, 1), St, (HW * HW, HW), N, HW, HW, Cc, alpha=float(int(Cc) ** (-0.5)))
        ops.softmax_c_(St, N, HW, HW)                  # softmax over keys
        # h[c][i] = sum_j v[c][j] * P[i][j]                                         (model.py:191-195)
        ho = torch.empty((N, Cc, H, W), dtype=torch.float32, device=qkv.device)
        ops.bgemm(v, (bs, HW, 1), St, (HW * HW, HW, 1), ho, (Cc * HW, HW), N, Cc, HW, HW)
        return ho

    def forward(self, x: Tensor, out: Optional[Tensor] = None) -> Tensor:
        N, Cc, H, W = x.shape
        HW = H * W
        h_ = self.norm(x)
        qkv = self._qkv()(h_)                          # [N, 3C, H, W]
        if HW % 64 == 0 and Cc in (128, 256, 512) and os.environ.get("DCVIC_ATTN_FUSED", "1") != "0":
            # flash-style fused kernel: the HW x HW score matrix never reaches HBM (csrc/attn.hip)
            ho = ops.attn_fused(qkv, Cc)
        else:
            ho = self._attn_unfused(qkv, N, Cc, H, W)
        return self.proj_out(ho, res=x, out=out)


class Downsample(nn.Module):
    def __init__(self, in_channels: int, with_conv: bool = True):
        super().__init__()
        assert with_conv
        self.conv = Conv2d(in_channels, in_channels, 3, stride=2, padding=0, asym_pad=True)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv(x)


class Upsample(nn.Module):
    def __init__(self, in_channels: int, with_conv: bool = True):
        super().__init__()
        assert with_conv
        self.conv = Conv2d(in_channels, in_channels, 3, 1, 1, upsample=True)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv(x)


class Encoder(nn.Module):
    """ldm model.py:368-459."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignore_kwargs):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = Conv2d(in_channels, ch, 3, 1, 1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = GroupNorm(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)
        if os.environ.get("DCVIC_WINO_ENCODER", "1") != "0":
            allow_winograd(self)

    def forward(self, x: Tensor) -> Tensor:
        h = self.conv_in(x)
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = lvl.block[i_block](h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = lvl.downsample(h)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        h = self.norm_out(h, act=ops.ACT_SWISH, out=h)
        return self.conv_out(h)


class Decoder(nn.Module):
    """ldm model.py:462-568 (module tree).  The forward used on the DC-VIC path is driven layer by
    layer from VqDecFusionModule (dc_vic_amd/fusion.py); `forward` here is the plain VQGAN decoder."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 **ignorekwargs):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        if tanh_out:
            raise NotImplementedError("tanh_out is not used by the shipped configs")
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = GroupNorm(block_in)
        self.conv_out = Conv2d(block_in, out_ch, 3, 1, 1)
        allow_winograd(self, f44=True)      # after the estimator argmax: only the reconstruction's fp tolerance depends on these layers

    def forward(self, z: Tensor) -> Tensor:
        self.last_z_shape = z.shape
        h = self.conv_in(z)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        part = None                                   # GroupNorm statistics of `h` from its producer, while `h` comes straight out of a ResnetBlock
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                blk = self.up[i_level].block[i_block]
                has_attn = len(self.up[i_level].attn) > 0
                h = blk(h, in_part=part, want_part=not has_attn)
                part = blk.out_part
                if has_attn:
                    h = self.up[i_level].attn[i_block](h)
                    part = None
            if i_level != 0:
                h = self.up[i_level].upsample(h)
                part = None
        if self.give_pre_end:
            return h
        h = self.norm_out(h, act=ops.ACT_SWISH, out=h, part=part)
        return self.conv_out(h)


class Embedding(nn.Module):
    def __init__(self, n: int, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d).uniform_(-1.0 / n, 1.0 / n), requires_grad=False)

    def forward(self, indices: Tensor) -> Tensor:
        # gather is only needed for API parity (vq_indices_to_latent); inference uses argmax_lut
        return self.weight[indices]


class VectorQuantizer2(nn.Module):
    """taming/modules/vqvae/quantize.py:213-329 (remap None, legacy True): forward returns
    (z_q, loss, (perplexity, min_encodings, indices)) like the reference; loss is None at inference."""

    def __init__(self, n_e: int, e_dim: int, beta: float = 0.25, sane_index_shape: bool = False, **kwargs):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = Embedding(n_e, e_dim)
        self.sane_index_shape = sane_index_shape

    def forward(self, z: Tensor, want_feat: bool = False):
        idx, zq, feat = ops.vq_argmin(z.contiguous(), self.embedding.weight, want_zq=True, want_feat=want_feat)
        if not self.sane_index_shape:
            idx = idx.reshape(-1, 1)
        if want_feat:
            return zq, None, (None, None, idx), feat
        return zq, None, (None, None, idx)


class VQModelInterface(nn.Module):
    """ldm/models/autoencoder.py:14-61, 264-282."""

    def __init__(self, embed_dim: int, ddconfig: dict, n_embed: int, lossconfig=None, ckpt_path=None, monitor=None, **kwargs):
        super().__init__()
        self.embed_dim, self.n_embed = embed_dim, n_embed
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quantize = VectorQuantizer2(n_embed, embed_dim, beta=0.25)
        self.quant_conv = Conv2d(ddconfig["z_channels"], embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)

    def encode(self, x: Tensor) -> Tensor:
        return self.quant_conv(self.encoder(x))

    def decode(self, h: Tensor, force_not_quantize: bool = False) -> Tensor:
        quant = h if force_not_quantize else self.quantize(h)[0]
        return self.decoder(self.post_quant_conv(quant))
