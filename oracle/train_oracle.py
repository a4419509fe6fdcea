"""CPU oracle of the stage-3 training step: torch autograd over the functional restatement in dcvic_oracle.py.

TEST INFRASTRUCTURE, NOT PRODUCT (same rule as the rest of oracle/).  Restates
  src/trainer/dual_cond_gan_distortion_vq_code_trainer.py:116-300 (run_comp_model, calc_g_loss, run_discriminator, calc_d_loss),
  src/models/comp_model/hyperprior_dc_vic_model.py:208-274 (forward, is_train=True, fix_entropy_models=True),
  src/models/discriminator/dual_beta_taming_nlayer_discriminator.py:71-89 + taming_nlayer_discriminator.py:65-119,
  src/losses/{distortion_loss.py:11-39, gan_loss.py:10-32, cross_entropy_loss.py:10-28} with config/exp1_stage1_3.yaml:61-79.
Pinning: the sub-network forwards are the golden-pinned functions of dcvic_oracle.py; gradients are torch.autograd's.
The trainer module itself cannot be imported (wandb / compressai / lpips) -> its step wiring is restated from the source
("parity unpinned" at the wiring level); LPIPS is excluded on both sides (weights cannot be fetched).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import dcvic_oracle as O

TRAINABLE_PREFIXES = ("decoder.", "vq_estimator.", "fusion_module.")
LOSS_W = dict(distortion=50.0, perceptual=1.0, gan=0.01, code_distortion=1.0, code_ce=0.5)


def discriminator(dsd: Dict[str, torch.Tensor], x: torch.Tensor, b1, b2) -> torch.Tensor:
    """DualBetaCondTamingNLayerDiscriminator.forward (n_layers 3, ndf 64, norm none, cond_ch 8, L 10, include_x, no pi)."""
    N, _, H, W = x.shape
    c = torch.cat([O.fourier_embed(b1, O.MAX_BETA_1), O.fourier_embed(b2, O.MAX_BETA_2)], dim=1)
    c = F.linear(F.relu(F.linear(c, dsd["mlp.0.weight"], dsd["mlp.0.bias"])), dsd["mlp.2.weight"], dsd["mlp.2.bias"])
    c = c.unsqueeze(-1).unsqueeze(-1).expand(N, -1, H, W)
    h = torch.cat([x, c], dim=1)
    for i, s in ((0, 2), (2, 2), (5, 2), (8, 1)):
        h = F.leaky_relu(F.conv2d(h, dsd[f"main.{i}.weight"], dsd[f"main.{i}.bias"], stride=s, padding=1), 0.2)
    return F.conv2d(h, dsd["main.11.weight"], dsd["main.11.bias"], stride=1, padding=1)


def lpips_alex(lsd: Dict[str, torch.Tensor], in0: torch.Tensor, in1: torch.Tensor) -> torch.Tensor:
    """lpips.LPIPS(net='alex', lpips=True, spatial=False).forward(in0, in1) -> [N, 1, 1, 1], restated from the published model
    (PARITY UNPINNED: package and weights absent).  `lsd`: state dict with the package's key names."""
    def feats(x):
        x = (x - lsd["scaling_layer.shift"]) / lsd["scaling_layer.scale"]
        f1 = F.relu(F.conv2d(x, lsd["net.slice1.0.weight"], lsd["net.slice1.0.bias"], stride=4, padding=2))
        f2 = F.relu(F.conv2d(F.max_pool2d(f1, 3, 2), lsd["net.slice2.3.weight"], lsd["net.slice2.3.bias"], padding=2))
        f3 = F.relu(F.conv2d(F.max_pool2d(f2, 3, 2), lsd["net.slice3.6.weight"], lsd["net.slice3.6.bias"], padding=1))
        f4 = F.relu(F.conv2d(f3, lsd["net.slice4.8.weight"], lsd["net.slice4.8.bias"], padding=1))
        f5 = F.relu(F.conv2d(f4, lsd["net.slice5.10.weight"], lsd["net.slice5.10.bias"], padding=1))
        return [f1, f2, f3, f4, f5]

    def unit(f):
        return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + 1e-10)
    val = 0
    for k, (a, b) in enumerate(zip(feats(in0), feats(in1))):
        d = (unit(a) - unit(b)) ** 2
        val = val + F.conv2d(d, lsd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return val


def encode_side(sd, x: torch.Tensor, b1, b2, eb):
    """Frozen part under no_grad: VQGAN encode + VQ, ELIC encoder, hyperprior, CHARM -> y_hat (values of ste_round = round)."""
    with torch.no_grad():
        z_q, idx = O.vq_encode(sd, x)
        y = O.elic_encoder(sd, x, O.onehot_feat(sd, z_q, idx), b1, b2)
        z = O.hyper_encoder(sd, y)
        z_hat, z_lik = eb.forward(z)
        ch = O.charm_forward(sd, y, O.hyper_decoder(sd, z_hat))
    return z_q, idx, ch["y_hat"], ch["y_likelihood"], z_lik


def generator_losses(sd, dsd, x: torch.Tensor, b1, b2, eb, w=LOSS_W, lsd=None, force_y_hat=None, force_out_idx=None):
    """calc_g_loss on run_comp_model's output.  `sd` entries under TRAINABLE_PREFIXES should require grad.
    `force_y_hat` / `force_out_idx` (test infrastructure): evaluate on given integer decisions (rounded symbols, estimator argmax)
    instead of this CPU evaluation's own -- the gradients are only comparable between two implementations that took the same ones."""
    z_q, idx, y_hat, _, _ = encode_side(sd, x, b1, b2, eb)
    if force_y_hat is not None:
        y_hat = force_y_hat
    feat_1, feats = O.elic_decoder_feats(sd, y_hat, b1, b2)
    pred_embed, logits = O.swin_estimator(sd, feat_1)
    out_idx = torch.argmax(logits, dim=1) if force_out_idx is None else force_out_idx
    lat = O._conv(sd, "vq_model.post_quant_conv", O.vq_indices_to_latent(sd, out_idx))
    fake = O.fusion_decode(sd, lat, feats, 1.0)
    L = {}
    L["distortion"] = w["distortion"] * F.mse_loss((x + 1.0) / 2.0, (fake + 1.0) / 2.0)
    if lsd is not None:
        L["perceptual"] = w["perceptual"] * torch.mean(lpips_alex(lsd, x, fake))
    g_fake = discriminator(dsd, fake, b1, b2)
    L["adv"] = w["gan"] * F.binary_cross_entropy_with_logits(g_fake, torch.ones_like(g_fake))
    L["code_distortion"] = w["code_distortion"] * F.mse_loss(z_q, pred_embed)
    L["code_ce"] = w["code_ce"] * F.cross_entropy(logits, idx)
    return L, dict(fake=fake, logits=logits, pred_embed=pred_embed, out_idx=out_idx, gt_idx=idx, y_hat=y_hat, z_q=z_q)


def discriminator_losses(dsd, real: torch.Tensor, fake: torch.Tensor, b1, b2):
    d_real, d_fake = discriminator(dsd, real, b1, b2), discriminator(dsd, fake.detach(), b1, b2)
    l_real = 0.5 * F.binary_cross_entropy_with_logits(d_real, torch.ones_like(d_real))
    l_fake = 0.5 * F.binary_cross_entropy_with_logits(d_fake, torch.zeros_like(d_fake))
    return l_real, l_fake, d_real, d_fake


def clip_and_adam(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], lr: float, max_norm=None, step: int = 1):
    """One torch-semantics Adam step from zero moments (+ clip_grad_norm_) -> new parameter values."""
    names = sorted(params)
    ps = [params[k].detach().clone().requires_grad_(True) for k in names]
    for p, k in zip(ps, names):
        p.grad = grads[k].clone()
    if max_norm:
        torch.nn.utils.clip_grad_norm_(ps, max_norm)
    opt = torch.optim.Adam(ps, lr=lr)
    opt.step()
    return {k: p.detach() for k, p in zip(names, ps)}
