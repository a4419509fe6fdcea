"""beta-conditioned ELIC encoder / decoder-feature network on HIP kernels.

Mirrors (module tree + arithmetic):
  src/models/subnet/autoencoder/elic_dual_beta_ft_autoencoder.py:27-45 (BetaScaleShiftModule),
  :48-141 (ElicDualBetaFtVqScEncoder), :226-359 (ElicDualBetaFtFeatFusionDecoder);
  src/models/subnet/autoencoder/elic_autoencoder.py:21-71; src/models/layer/elic_layers.py:15-45;
  src/models/layer/cheng_nlam.py:5-47; src/models/layer/fourier_enc.py:10-41.
The beta conditioning (Fourier features -> MLP -> per-layer scale/shift vectors) depends only on
(beta_rate, beta_vq); for scalar betas the vectors are computed once and cached, and the affine
`x*(1+s)+t` is fused into the epilogue of the producing convolution wherever one exists.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .layers import Act, Conv2d, ConvTranspose2d, Linear
from .registry import DECODER_REGISTRY, ENCODER_REGISTRY

Tensor = torch.Tensor


class FourierEncoding:
    """fourier_enc.py:10-41 (host-side, as in the reference: beta.cpu())."""

    def __init__(self, L: int, max_beta: float, use_pi: bool = True, include_x: bool = False):
        assert L > 0 and max_beta > 0
        self.L, self.max_beta = L, max_beta
        self.freq = torch.pow(torch.Tensor([2]), torch.arange(L)).unsqueeze(0)
        if use_pi:
            self.freq = self.freq * np.pi
        self.include_x = include_x

    def embed(self, beta: Union[int, float, Tensor]) -> Tensor:
        if isinstance(beta, (int, float)):
            beta = torch.Tensor([beta]).float()
        assert isinstance(beta, Tensor) and beta.ndim == 1
        assert 0 <= beta.min() <= self.max_beta and 0 <= beta.max() <= self.max_beta
        beta = beta.detach().cpu().float()
        nb = ((beta / self.max_beta) - 0.5) * 2
        nb = nb.unsqueeze(1)
        out = torch.cat([torch.sin(nb * self.freq), torch.cos(nb * self.freq)], dim=-1)
        if self.include_x:
            out = torch.cat([nb, out], dim=-1)
        return out


class BaseBlock(nn.Module):
    """elic_layers.py:15-30: 1x1 -> ReLU -> 3x3 -> ReLU -> 1x1, + skip."""

    def __init__(self, ch: int, mid_ch: int):
        super().__init__()
        self.conv = nn.Sequential(Conv2d(ch, mid_ch, 1), Act(), Conv2d(mid_ch, mid_ch, 3, 1, 1), Act(), Conv2d(mid_ch, ch, 1))

    def forward(self, x: Tensor, affine=None, out=None) -> Tensor:
        h = self.conv[0](x, act=ops.ACT_RELU)
        h = self.conv[2](h, act=ops.ACT_RELU)
        return self.conv[4](h, res=x, affine=affine, out=out)


class ResidualBottleneckBlocks(nn.Module):
    def __init__(self, ch: int, mid_ch: int, num_blocks: int = 3, res_in_res: bool = False):
        super().__init__()
        assert not res_in_res
        self.num_blocks = num_blocks
        for i in range(num_blocks):
            setattr(self, f"block{i}", BaseBlock(ch, mid_ch))

    def forward(self, x: Tensor, affine=None, out=None) -> Tensor:
        y = x
        for i in range(self.num_blocks):
            last = i == self.num_blocks - 1
            y = getattr(self, f"block{i}")(y, affine=affine if last else None, out=out if last else None)
        return y


class NLAMResBlock(nn.Module):
    """cheng_nlam.py:30-47."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        mid = out_ch // 2
        self.c1 = Conv2d(in_ch, mid, 1)
        self.c2 = Conv2d(mid, mid, 3, 1, 1)
        self.c3 = Conv2d(mid, out_ch, 1)

    def forward(self, x: Tensor) -> Tensor:
        o = self.c1(x, act=ops.ACT_RELU)
        o = self.c2(o, act=ops.ACT_RELU)
        return self.c3(o, res=x)


class ChengNLAM(nn.Module):
    """cheng_nlam.py:5-28: x + trunk(x) * sigmoid(conv1x1(attention(x)))."""

    def __init__(self, ch: int):
        super().__init__()
        self.trunk_block = nn.Sequential(*[NLAMResBlock(ch, ch) for _ in range(3)])
        self.attention_block = nn.Sequential(*[NLAMResBlock(ch, ch) for _ in range(3)])
        self.conv = Conv2d(ch, ch, 1)

    def forward(self, x: Tensor) -> Tensor:
        t, a = x, x
        for i in range(3):
            t = self.trunk_block[i](t)
            a = self.attention_block[i](a)
        a = self.conv(a)
        return ops.add_mul_sigmoid(x, t, a)


class BetaScaleShiftModule(nn.Module):
    """elic_dual_beta_ft_autoencoder.py:27-45."""

    def __init__(self, cond_ch: int, feat_ch: int):
        super().__init__()
        self.shared = nn.Sequential(Conv2d(cond_ch, cond_ch, 1), Act())
        self.scale = Conv2d(cond_ch, feat_ch, 1)
        self.shift = Conv2d(cond_ch, feat_ch, 1)

    def vectors(self, cond: Tensor) -> Tuple[Tensor, Tensor]:
        """cond [B, cond_ch, 1, 1] -> contiguous (scale, shift) [B, feat_ch]."""
        c = self.shared[0](cond, act=ops.ACT_RELU)
        B = cond.shape[0]
        return self.scale(c).reshape(B, -1).contiguous(), self.shift(c).reshape(B, -1).contiguous()

    def forward(self, feat: Tensor, cond: Tensor) -> Tensor:
        s, t = self.vectors(cond)
        return ops.chan_affine(feat, s, t)


class _BetaCond(nn.Module):
    """Shared by encoder and decoder: embed_1/embed_2 + mlp -> cond, per-layer vectors cached per beta pair."""

    def _init_cond(self, cond_ch, L, use_pi, include_x, max_beta_1, max_beta_2):
        self.embed_1 = FourierEncoding(L=L, max_beta=max_beta_1, use_pi=use_pi, include_x=include_x)
        self.embed_2 = FourierEncoding(L=L, max_beta=max_beta_2, use_pi=use_pi, include_x=include_x)
        mlp_in = 2 * (2 * L + 1) if include_x else 2 * 2 * L
        self.mlp = nn.Sequential(Linear(mlp_in, cond_ch), Act(), Linear(cond_ch, cond_ch))
        self._vec_cache: Dict = {}

    def cond(self, beta_1, beta_2, device) -> Tensor:
        c = torch.cat([self.embed_1.embed(beta_1), self.embed_2.embed(beta_2)], dim=1).to(device)  # [B, 4L(+2)]
        c = c.reshape(c.shape[0], -1, 1, 1).contiguous()
        c = self.mlp[0](c, act=ops.ACT_RELU)
        return self.mlp[2](c)                                                                      # [B, cond_ch, 1, 1]

    def beta_vectors(self, beta_1, beta_2, device, modules: List[BetaScaleShiftModule]):
        scalar = isinstance(beta_1, (int, float)) and isinstance(beta_2, (int, float))
        key = None
        if scalar:
            key = (float(beta_1), float(beta_2), str(device))
            hit = self._vec_cache.get(key)
            if hit is not None:
                return hit
        c = self.cond(beta_1, beta_2, device)
        vecs = [m.vectors(c) for m in modules]
        if scalar:
            if len(self._vec_cache) > 16:
                self._vec_cache.clear()
            self._vec_cache[key] = vecs
        return vecs

    def invalidate_caches(self):
        """Called after weights change (load_state_dict / .to()): the cached vectors depend on them."""
        self._vec_cache.clear()

    def _apply(self, fn, *args, **kwargs):
        if hasattr(self, "_vec_cache"):
            self._vec_cache.clear()
        return super()._apply(fn, *args, **kwargs)


@ENCODER_REGISTRY.register()
class ElicDualBetaFtVqScEncoder(_BetaCond):
    """elic_dual_beta_ft_autoencoder.py:48-141."""

    def __init__(self, in_ch: int = 3, out_ch: int = 192, main_ch: int = 192, block_mid_ch: int = 192, num_blocks: int = 3,
                 max_beta_1: float = 5.12, max_beta_2: float = 5.12, cond_ch: int = 512, L: int = 10, use_pi: bool = True,
                 include_x: bool = False, input_feat_ch: int = 5, proj_init: bool = True, proj_init_std: float = 0.02):
        super().__init__()
        self.conv1 = Conv2d(in_ch, main_ch, 5, 2, 2)
        self.block1 = ResidualBottleneckBlocks(main_ch, block_mid_ch, num_blocks)
        self.conv2 = Conv2d(main_ch, main_ch, 5, 2, 2)
        self.block2 = ResidualBottleneckBlocks(main_ch, block_mid_ch, num_blocks)
        self.attn2 = ChengNLAM(main_ch)
        self.conv3 = Conv2d(main_ch, main_ch, 5, 2, 2)
        self.block3 = ResidualBottleneckBlocks(main_ch, block_mid_ch, num_blocks)
        self.conv4 = Conv2d(main_ch, out_ch, 5, 2, 2)
        self.attn4 = ChengNLAM(out_ch)
        self.num_downscale = 4
        chs = [main_ch] * 7 + [out_ch] * 2
        self.beta_ft_list = nn.ModuleList([BetaScaleShiftModule(cond_ch, c) for c in chs])
        self._init_cond(cond_ch, L, use_pi, include_x, max_beta_1, max_beta_2)
        self.projection = Conv2d(main_ch + input_feat_ch, main_ch, 3, 1, 1)
        self.input_vq_latent = True

    def forward(self, x: Tensor, feat: Tensor, beta_1, beta_2) -> Tensor:
        v = self.beta_vectors(beta_1, beta_2, x.device, list(self.beta_ft_list))
        x = self.conv1(x, affine=v[0])
        x = self.block1(x, affine=v[1])
        x = self.conv2(x, affine=v[2])
        x = self.block2(x, affine=v[3])
        x = ops.chan_affine(self.attn2(x), *v[4])
        x = self.conv3(x, affine=v[5])
        x = self.projection([feat, x], res=x)          # x + conv3x3(cat[feat, x])   (:131-132)
        x = self.block3(x, affine=v[6])
        x = self.conv4(x, affine=v[7])
        x = ops.chan_affine(self.attn4(x), *v[8])
        return x


@DECODER_REGISTRY.register()
class ElicDualBetaFtFeatFusionDecoder(_BetaCond):
    """elic_dual_beta_ft_autoencoder.py:226-359 (get_feats only; forward raises like the reference)."""

    def __init__(self, fusion_layer_dict: Dict[str, str], feat_layer_name: str, in_ch: int = 192, out_ch: int = 3,
                 main_ch: int = 192, block_mid_ch: int = 192, num_blocks: int = 3, use_tanh: bool = True,
                 pixel_shuffle: bool = False, res_in_res: bool = False, max_beta_1: float = 5.12, max_beta_2: float = 5.12,
                 cond_ch: int = 512, L: int = 10, use_pi: bool = True, include_x: bool = False,
                 beta_weight_init: bool = False, beta_weight_init_std: float = 0.02):
        super().__init__()
        assert not pixel_shuffle and not res_in_res
        self.use_tanh = use_tanh
        up = lambda i, o: ConvTranspose2d(i, o, 5, 2, 2, 1)
        self.attn1 = ChengNLAM(in_ch)
        self.conv1 = up(in_ch, main_ch)
        self.block1 = ResidualBottleneckBlocks(main_ch, block_mid_ch, num_blocks)
        self.conv2 = up(main_ch, main_ch)
        self.attn2 = ChengNLAM(main_ch)
        self.block2 = ResidualBottleneckBlocks(main_ch, block_mid_ch, num_blocks)
        self.conv3 = up(main_ch, main_ch)
        self.block3 = ResidualBottleneckBlocks(main_ch, block_mid_ch, num_blocks)
        self.conv4 = up(main_ch, out_ch)     # present in checkpoints, never executed (loop breaks first, :356-357)
        self.layer_names = ["attn1", "conv1", "block1", "conv2", "attn2", "block2", "conv3", "block3", "conv4"]
        self.feat_layer = feat_layer_name
        assert self.feat_layer in self.layer_names
        self.fusion_layer_dict = dict(fusion_layer_dict)
        for k in self.fusion_layer_dict:
            assert k in self.layer_names
        self.beta_ft_list = nn.ModuleList([BetaScaleShiftModule(cond_ch, c) for c in [in_ch, in_ch] + [main_ch] * 7])
        self.max_beta_1, self.max_beta_2 = max_beta_1, max_beta_2
        self._init_cond(cond_ch, L, use_pi, include_x, max_beta_1, max_beta_2)
        self.init_fuse = BetaScaleShiftModule(cond_ch, main_ch)

    def forward(self, x):
        raise NotImplementedError()

    def get_feats(self, x: Tensor, beta_1, beta_2, feat_out: Optional[Dict[str, Tensor]] = None):
        """Returns (feat_1, {fusion key: feature}).  `feat_out` may map fusion keys to pre-allocated
        destination views (e.g. the first 192 channels of a fusion block's concat buffer)."""
        mods = [self.init_fuse] + list(self.beta_ft_list)
        v = self.beta_vectors(beta_1, beta_2, x.device, mods)
        x = ops.chan_affine(x, *v[0], add_=x)                       # init_fuse(x, c) + x   (:343)
        fusion_feat_dict: Dict[str, Tensor] = {}
        query = list(self.fusion_layer_dict.keys())
        feat_1 = None
        for li, name in enumerate(self.layer_names):
            layer = getattr(self, name)
            x = ops.chan_affine(x, *v[1 + li])
            dst = None
            if feat_out is not None and name in self.fusion_layer_dict:
                dst = feat_out.get(self.fusion_layer_dict[name])
            if isinstance(layer, ResidualBottleneckBlocks):
                x = layer(x, out=dst)
            else:
                x = layer(x)
            if name == self.feat_layer:
                feat_1 = x
            if name in query:
                fusion_feat_dict[self.fusion_layer_dict[name]] = x
            if len(fusion_feat_dict) == len(query):
                break
        return feat_1, fusion_feat_dict
