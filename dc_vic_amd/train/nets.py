"""Training-mode (differentiable) forwards of the sub-networks the stage-3 trainer optimises, written against the SAME
parameter-holding modules the inference path uses (dc_vic_amd/{elic,swin,fusion,vqgan}.py), plus the PatchGAN
discriminator.  Reference: hyperprior_dc_vic_model.py:208-274 (forward, is_train=True, fix_entropy_models=True),
elic_dual_beta_ft_autoencoder.py:332-359, swin_vq_estimator.py:70-98, vq_fusion_module.py:78-126 driving ldm
model.py:462-568, dual_beta_taming_nlayer_discriminator.py:16-89 / taming_nlayer_discriminator.py:29-119.
Only decoder, vq_estimator and fusion_module are trainable (dual_cond_gan_distortion_vq_code_trainer.py:47-52); the frozen
VQGAN decoder is differentiated through (data gradients only)."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..elic import BetaScaleShiftModule, ChengNLAM, ConvTranspose2d, FourierEncoding, ResidualBottleneckBlocks
from ..layers import Act, Conv2d, Linear
from . import autograd as A
from .autograd import Ctx, Var

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------- beta conditioning
def cond_vector(ctx: Ctx, owner, beta_1, beta_2, device) -> Var:
    """embed_1 / embed_2 (host, constants) -> mlp -> cond [B, cond_ch, 1, 1]   (elic_dual_beta_ft_autoencoder.py:109-113)."""
    c = torch.cat([owner.embed_1.embed(beta_1), owner.embed_2.embed(beta_2)], dim=1).to(device)
    c = c.reshape(c.shape[0], -1, 1, 1).contiguous().float()
    h = A.conv(ctx, A.const(c), owner.mlp[0], act=ops.ACT_RELU)
    return A.conv(ctx, h, owner.mlp[2])


def beta_vectors(ctx: Ctx, m: BetaScaleShiftModule, cond: Var) -> Tuple[Var, Var]:
    c = A.conv(ctx, cond, m.shared[0], act=ops.ACT_RELU)
    return A.conv(ctx, c, m.scale), A.conv(ctx, c, m.shift)


# ------------------------------------------------------------------------------------------- ELIC blocks
def bottleneck_blocks(ctx: Ctx, m: ResidualBottleneckBlocks, x: Var) -> Var:
    y = x
    for i in range(m.num_blocks):
        b = getattr(m, f"block{i}")
        h = A.conv(ctx, y, b.conv[0], act=ops.ACT_RELU)
        h = A.conv(ctx, h, b.conv[2], act=ops.ACT_RELU)
        y = A.add(ctx, A.conv(ctx, h, b.conv[4]), y)
    return y


def _nlam_res(ctx: Ctx, b, x: Var) -> Var:
    o = A.conv(ctx, x, b.c1, act=ops.ACT_RELU)
    o = A.conv(ctx, o, b.c2, act=ops.ACT_RELU)
    return A.add(ctx, A.conv(ctx, o, b.c3), x)


def nlam(ctx: Ctx, m: ChengNLAM, x: Var) -> Var:
    t, a = x, x
    for i in range(3):
        t = _nlam_res(ctx, m.trunk_block[i], t)
        a = _nlam_res(ctx, m.attention_block[i], a)
    return A.nlam_gate(ctx, x, t, A.conv(ctx, a, m.conv))


def decoder_get_feats(ctx: Ctx, dec, y_hat: Tensor, beta_1, beta_2):
    """ElicDualBetaFtFeatFusionDecoder.get_feats with a tape (y_hat is a constant: the entropy side runs under no_grad)."""
    cond = cond_vector(ctx, dec, beta_1, beta_2, y_hat.device)
    x = A.const(y_hat)
    s0, t0 = beta_vectors(ctx, dec.init_fuse, cond)
    x = A.chan_affine(ctx, x, s0, t0, add_x=True)                         # init_fuse(x, c) + x   (:343)
    feats: Dict[str, Var] = {}
    query = list(dec.fusion_layer_dict.keys())
    feat_1 = None
    for li, name in enumerate(dec.layer_names):
        layer = getattr(dec, name)
        s, t = beta_vectors(ctx, dec.beta_ft_list[li], cond)
        x = A.chan_affine(ctx, x, s, t)
        if isinstance(layer, ResidualBottleneckBlocks):
            x = bottleneck_blocks(ctx, layer, x)
        elif isinstance(layer, ChengNLAM):
            x = nlam(ctx, layer, x)
        else:
            x = A.conv(ctx, x, layer)
        if name == dec.feat_layer:
            feat_1 = x
        if name in query:
            feats[dec.fusion_layer_dict[name]] = x
        if len(feats) == len(query):
            break
    return feat_1, feats


# ------------------------------------------------------------------------------------------- Swin estimator
def _femasr_res(ctx: Ctx, rb, x: Var) -> Var:
    h = A.group_norm(ctx, x, rb.conv[0].norm, act=ops.ACT_SWISH)
    h = A.conv(ctx, h, rb.conv[2])
    h = A.group_norm(ctx, h, rb.conv[3].norm, act=ops.ACT_SWISH)
    return A.add(ctx, A.conv(ctx, h, rb.conv[5]), x)


def _swin_block(ctx: Ctx, blk, x: Var) -> Var:
    h = A.layer_norm_c(ctx, x, blk.norm1)
    qkv = A.conv(ctx, h, blk.attn.qkv)
    a = A.swin_attention(ctx, qkv, blk.attn.relative_position_bias_table, blk.num_heads, blk.window_size, blk.shift_size)
    x = A.add(ctx, A.conv(ctx, a, blk.attn.proj), x)
    h = A.layer_norm_c(ctx, x, blk.norm2)
    h = A.activation(ctx, A.conv(ctx, h, blk.mlp.fc1), ops.ACT_GELU)
    return A.add(ctx, A.conv(ctx, h, blk.mlp.fc2), x)


def estimator_forward(ctx: Ctx, est, feat: Var) -> Tuple[Var, Var]:
    """DualBlockSwinVqEstimator.forward -> (pred_embed, logits)."""
    x = A.conv(ctx, feat, est.first_block[0])
    x = _femasr_res(ctx, est.first_block[2], x)
    x = _femasr_res(ctx, est.first_block[3], x)
    x = A.conv(ctx, x, est.first_block[4])
    pred_embed = A.conv(ctx, x, est.embed_projection)
    h, w = x.shape[2:]
    if h % est.window_size or w % est.window_size:
        raise NotImplementedError("training crops are 256x256 (32x32 tokens): the reflect-padded estimator path is inference-only")
    for rstb in est.swin_blks:
        g = x
        for blk in rstb.residual_group.blocks:
            g = _swin_block(ctx, blk, g)
        x = A.add(ctx, A.conv(ctx, g, rstb.conv), x)
    x = _femasr_res(ctx, est.out_block[0], x)
    return pred_embed, A.conv(ctx, x, est.out_block[1])


# ------------------------------------------------------------------------------------------- VQGAN decoder + SFT fusion
def _ldm_resnet(ctx: Ctx, rb, x: Var) -> Var:
    h = A.group_norm(ctx, x, rb.norm1, act=ops.ACT_SWISH)
    h = A.conv(ctx, h, rb.conv1)
    h = A.group_norm(ctx, h, rb.norm2, act=ops.ACT_SWISH)
    h = A.conv(ctx, h, rb.conv2)
    skip = A.conv(ctx, x, rb.nin_shortcut) if rb.in_channels != rb.out_channels else x
    return A.add(ctx, skip, h)


def _ldm_attn(ctx: Ctx, ab, x: Var) -> Var:
    h = A.group_norm(ctx, x, ab.norm)
    qkv = A.cat_channels(ctx, [A.conv(ctx, h, ab.q), A.conv(ctx, h, ab.k), A.conv(ctx, h, ab.v)])
    o = A.attn_single_head(ctx, qkv, ab.in_channels)
    return A.add(ctx, x, A.conv(ctx, o, ab.proj_out))


def _cf_resblock(ctx: Ctx, rb, x_in: Var) -> Var:
    x = A.group_norm(ctx, x_in, rb.norm1, act=ops.ACT_SWISH)
    x = A.conv(ctx, x, rb.conv1)
    x = A.group_norm(ctx, x, rb.norm2, act=ops.ACT_SWISH)
    x = A.conv(ctx, x, rb.conv2)
    skip = A.conv(ctx, x_in, rb.conv_out) if rb.in_channels != rb.out_channels else x_in
    return A.add(ctx, x, skip)


def _fuse_sft(ctx: Ctx, fb, dec: Var, cond: Var, w: float) -> Var:
    f = _cf_resblock(ctx, fb.fuse_block, A.cat_channels(ctx, [cond, dec]))
    sc = A.conv(ctx, A.conv(ctx, f, fb.scale[0], act=ops.ACT_LRELU02), fb.scale[2])
    sh = A.conv(ctx, A.conv(ctx, f, fb.shift[0], act=ops.ACT_LRELU02), fb.shift[2])
    return A.sft(ctx, dec, sc, sh, w)


def fusion_decode(ctx: Ctx, fm, vq_dec, z: Var, cond_feats: Dict[str, Var], w: float = 1.0) -> Var:
    """VqDecFusionModule.forward (vq_fusion_module.py:78-126) with a tape."""
    h = A.conv(ctx, z, vq_dec.conv_in)
    h = _ldm_resnet(ctx, vq_dec.mid.block_1, h)
    h = _ldm_attn(ctx, vq_dec.mid.attn_1, h)
    h = _ldm_resnet(ctx, vq_dec.mid.block_2, h)
    for i_level in reversed(range(vq_dec.num_resolutions)):
        lvl = vq_dec.up[i_level]
        for i_block in range(vq_dec.num_res_blocks + 1):
            h = _ldm_resnet(ctx, lvl.block[i_block], h)
            if len(lvl.attn) > 0:
                h = _ldm_attn(ctx, lvl.attn[i_block], h)
        key = f"block_1_{2 ** i_level}"
        if key in fm.fusion_keys:
            h = _fuse_sft(ctx, fm.fusion_modules[key], h, cond_feats[key], w)
        if i_level != 0:
            h = A.conv(ctx, h, lvl.upsample.conv)
    h = A.group_norm(ctx, h, vq_dec.norm_out, act=ops.ACT_SWISH)
    return A.conv(ctx, h, vq_dec.conv_out)


# ------------------------------------------------------------------------------------------- discriminator
class DualBetaCondTamingNLayerDiscriminator(nn.Module):
    """PatchGAN conditioned on (beta_rate, beta_vq): state-dict keys `main.{0,2,5,8,11}.{weight,bias}`, `mlp.{0,2}.{weight,bias}`
    as in the reference (norm_type 'none' -> Identity layers keep their Sequential indices, biases on)."""

    def __init__(self, input_nc: int = 11, ndf: int = 64, out_nc: int = 1, n_layers: int = 3, keep_shape: bool = False,
                 use_actnorm: bool = False, norm_type: str = "none", norm_kwargs: dict = {}, max_beta_1: float = -1.0,
                 max_beta_2: float = -1.0, L: int = 10, cond_ch: int = 8, use_pi: bool = False, include_x: bool = True,
                 y_hat_cond: bool = False, y_hat_in_ch=None, y_hat_out_ch=None, weight_init: bool = True, **kwargs):
        super().__init__()
        if norm_type != "none" or use_actnorm or y_hat_cond or n_layers != 3:
            raise NotImplementedError("only the shipped PatchGAN (n_layers 3, norm_type none, no y_hat condition) is built")
        kw = 4
        seq = [Conv2d(input_nc, ndf, kw, 2, 1), Act()]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [Conv2d(ndf * prev, ndf * mult, kw, 2, 1), nn.Identity(), Act()]
        kl = 3 if keep_shape else kw
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [Conv2d(ndf * prev, ndf * mult, kl, 1, 1), nn.Identity(), Act()]
        seq += [Conv2d(ndf * mult, out_nc, kl, 1, 1)]
        self.main = nn.Sequential(*seq)
        self.embed_1 = FourierEncoding(L=L, max_beta=max_beta_1, use_pi=use_pi, include_x=include_x)
        self.embed_2 = FourierEncoding(L=L, max_beta=max_beta_2, use_pi=use_pi, include_x=include_x)
        mlp_in = 2 * (2 * L + 1) if include_x else 2 * 2 * L
        self.mlp = nn.Sequential(Linear(mlp_in, cond_ch), Act(), Linear(cond_ch, cond_ch))
        self.cond_ch = cond_ch
        self.y_hat_cond = False
        if weight_init:         # taming weights_init: conv weights ~ N(0, 0.02)
            for m in self.main:
                if isinstance(m, Conv2d):
                    nn.init.normal_(m.weight, 0.0, 0.02)


def discriminator_forward(ctx: Ctx, D: DualBetaCondTamingNLayerDiscriminator, x: Var, beta_1, beta_2) -> Var:
    N, _, H, W = x.shape
    dev = x.data.device
    cond = cond_vector(ctx, D, beta_1, beta_2, dev)                       # [B, cond_ch, 1, 1]
    zeros = torch.zeros((N, D.cond_ch, H, W), dtype=torch.float32, device=dev)
    zvec = torch.zeros((cond.shape[0], D.cond_ch, 1, 1), dtype=torch.float32, device=dev)
    cmap = A.chan_affine(ctx, A.const(zeros), A.const(zvec), cond)        # 0 * (1 + 0) + cond[n][c]: the broadcast, differentiable
    h = A.cat_channels(ctx, [x, cmap])
    h = A.conv(ctx, h, D.main[0], act=ops.ACT_LRELU02)
    h = A.conv(ctx, h, D.main[2], act=ops.ACT_LRELU02)
    h = A.conv(ctx, h, D.main[5], act=ops.ACT_LRELU02)
    h = A.conv(ctx, h, D.main[8], act=ops.ACT_LRELU02)
    return A.conv(ctx, h, D.main[11])
