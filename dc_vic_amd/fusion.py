"""SFT fusion of compression features into the frozen VQGAN decoder, on HIP kernels.

Mirrors src/models/subnet/vq_fusion_module.py:21-126 (VqDecFusionModule.forward driving the ldm
Decoder layer by layer) and src/models/layer/codeformer_layers.py:14-67 (ResBlock, FuseSftBlock).
torch.cat([cond, dec]) is never materialised: GroupNorm statistics need the concatenated channel
groups, so cond and dec are written side by side into one buffer by their producers (the ELIC
decoder tap and the last VQGAN block of the level) and read in place.
`forward_split` (vq_fusion_module.py:129-311) is unreachable for the shipped paths and
is not built; inputs that would enter it raise.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .layers import Act, Conv2d, GroupNorm, allow_winograd
from .registry import VQ_FUSION_REGISTRY

Tensor = torch.Tensor


def build_vq_fusion_module(vq_fusion_opt: Dict) -> nn.Module:
    opt = deepcopy(dict(vq_fusion_opt))
    network_type = opt.pop("type", "VqDecFusionModule")
    return VQ_FUSION_REGISTRY.get(network_type)(**opt)


class ResBlock(nn.Module):
    """codeformer_layers.py:20-43."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = GroupNorm(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = GroupNorm(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.conv_out = Conv2d(in_channels, out_channels, 1)

    def forward(self, x_in: Tensor) -> Tensor:
        x = self.norm1(x_in, act=ops.ACT_SWISH)
        x = self.conv1(x, gn_stats=True)              # (GroupNorm statistics from the F(4x4) epilogue: layers.Conv2d.forward)
        x = self.norm2(x, act=ops.ACT_SWISH, out=x, part=self.conv1.take_gn_part())
        skip = self.conv_out(x_in) if self.in_channels != self.out_channels else x_in
        return self.conv2(x, res=skip)


class FuseSftBlock(nn.Module):
    """codeformer_layers.py:46-67."""

    def __init__(self, cond_ch: int, dec_ch: int, mid_ch: int):
        super().__init__()
        self.cond_ch, self.dec_ch = cond_ch, dec_ch
        self.fuse_block = ResBlock(cond_ch + dec_ch, mid_ch)
        self.scale = nn.Sequential(Conv2d(mid_ch, dec_ch, 3, 1, 1), Act(), Conv2d(dec_ch, dec_ch, 3, 1, 1))
        self.shift = nn.Sequential(Conv2d(mid_ch, dec_ch, 3, 1, 1), Act(), Conv2d(dec_ch, dec_ch, 3, 1, 1))

    def forward(self, cat_buf: Tensor, w: float = 1.0) -> Tensor:
        """cat_buf: [N, cond_ch + dec_ch, H, W] holding cat[cond, dec] (written in place by the producers)."""
        dec = cat_buf[:, self.cond_ch:]
        f = self.fuse_block(cat_buf)
        sc = self.scale[2](self.scale[0](f, act=ops.ACT_LRELU02))
        sh = self.shift[2](self.shift[0](f, act=ops.ACT_LRELU02))
        return ops.sft(dec, sc, sh, w=w)


@VQ_FUSION_REGISTRY.register()
class VqDecFusionModule(nn.Module):
    def __init__(self, fuse_scedule_dict: Dict[str, dict], fuse_type: str = "sft", weight_init: bool = False, weight_init_std: float = 0.02):
        super().__init__()
        assert fuse_type == "sft", "only the 'sft' fusion of the shipped configs is built"
        for k, v in fuse_scedule_dict.items():
            assert isinstance(v, dict) and "cond_ch" in v and "dec_ch" in v
        self.fusion_modules = nn.ModuleDict({k: FuseSftBlock(v["cond_ch"], v["dec_ch"], v["mid_ch"]) for k, v in fuse_scedule_dict.items()})
        self.fusion_keys = list(fuse_scedule_dict.keys())
        allow_winograd(self, f44=True)      # decoder side, after the last integer decision (see layers.allow_winograd)
        for k in self.fusion_keys:
            assert k.startswith("block_1_"), "before_mid / after_mid fusion points are not used by the shipped configs"

    def alloc_cat_buffers(self, N: int, zH: int, zW: int, device) -> Dict[str, Tensor]:
        """One [N, cond+dec, H, W] buffer per fusion level; level block_1_s lives at (zH*8/s, zW*8/s)."""
        out = {}
        for k in self.fusion_keys:
            s = int(k.rsplit("_", 1)[1])
            m = self.fusion_modules[k]
            out[k] = torch.empty((N, m.cond_ch + m.dec_ch, zH * 8 // s, zW * 8 // s), dtype=torch.float32, device=device)
        return out

    def forward(self, z: Tensor, cond_feats: Dict[str, Tensor], vq_dec, w: float = 1.0, cat_bufs: Optional[Dict[str, Tensor]] = None) -> Tensor:
        """vq_fusion_module.py:78-126.  If `cat_bufs` is given, cond features already sit in the first
        cond_ch channels of each buffer; otherwise they are copied there."""
        N, _, H, W = z.shape
        if min(H * 8, W * 8) > 1024:
            raise NotImplementedError("forward_split is unreachable for the shipped paths (decode_split tiles first)")
        vq_dec.last_z_shape = z.shape
        if cat_bufs is None:
            cat_bufs = self.alloc_cat_buffers(N, H, W, z.device)
            for k in self.fusion_keys:
                c = cond_feats[k]
                ops.copy_planes(cat_bufs[k][:, : c.shape[1]], c, c.shape[2], c.shape[3])
        h = vq_dec.conv_in(z)
        h = vq_dec.mid.block_1(h)
        h = vq_dec.mid.attn_1(h)
        h = vq_dec.mid.block_2(h)
        part = None
        for i_level in reversed(range(vq_dec.num_resolutions)):
            lvl = vq_dec.up[i_level]
            key = f"block_1_{2 ** i_level}"
            fuse = key in self.fusion_keys
            nblk = vq_dec.num_res_blocks + 1
            for i_block in range(nblk):
                last = i_block == nblk - 1
                has_attn = len(lvl.attn) > 0
                dst = None
                if fuse and last:
                    m = self.fusion_modules[key]
                    dst = cat_bufs[key][:, m.cond_ch:]
                # `part`: GroupNorm statistics of h from the F(4x4) epilogue of the block that produced it, while the next consumer is a
                # GroupNorm over exactly that map (not after attention / SFT fusion / upsampling, not into a concat buffer)
                to_cat = dst is not None and not has_attn
                h = lvl.block[i_block](h, out=None if has_attn else dst, in_part=part, want_part=not has_attn and not to_cat)
                part = None if (has_attn or to_cat) else lvl.block[i_block].out_part
                if has_attn:
                    h = lvl.attn[i_block](h, out=dst)
            if fuse:
                h = self.fusion_modules[key](cat_bufs[key], w)
                part = None
            if i_level != 0:
                h = lvl.upsample(h)
                part = None
        if vq_dec.give_pre_end:
            return h
        h = vq_dec.norm_out(h, act=ops.ACT_SWISH, out=h, part=part)
        return vq_dec.conv_out(h)
