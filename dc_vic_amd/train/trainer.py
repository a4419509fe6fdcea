"""Stage-3 GAN training step of DC-VIC on the HIP kernels, data-parallel over one process per GPU.

Mirrors src/trainer/dual_cond_gan_distortion_vq_code_trainer.py:24-300 (`DualBetaCondGanDistortionVqCodeTrainer`):
`optimize_parameters` = G step (run_comp_model with is_train=True / fix_entropy_models=True / sample_batch_beta, calc_g_loss,
backward, clip_grad_norm_, Adam, MultiStepLR) followed by the D step (run_discriminator, calc_d_loss, backward, Adam).
Loss definitions: config/exp1_stage1_3.yaml:61-79 (MSELoss x50 on [0,1] images, VanillaGANLoss x0.01, VanillaMSELoss x1 on
the predicted code embedding, CrossEntropyLoss x0.5 on the code logits).

Differences, stated:
  * LPIPS (perceptual_loss): the `lpips` wheel and its AlexNet / head weights cannot be fetched offline.  The term is computed
    by dc_vic_amd/train/lpips.py (architecture restated, state-dict key names of the package) with deterministic synthetic
    weights unless an `lpips` state dict is loaded into `trainer.lpips` -- parity unpinned.  Pass `loss_weights={'perceptual': 0}`
    to drop the term.
  * the reference is single-process (README.md:64-65, base_trainer.py:158 TODO); here gradients are averaged across ranks
    with bucketed RCCL all-reduces of the flat gradient buffers -- with world size 1 nothing is sent.
Training mode changes nothing numerically on the frozen encoder / entropy side for this trainer: with
fix_entropy_models=True it runs under no_grad and ste_round(y - mu) + mu has the values of round(y - mu) + mu
(ste_gaussian_conditional.py:16-23), so the inference kernels provide y_hat; the noise-quantised likelihoods only feed a
rate loss this trainer does not have (the logged `qbpp` uses the quantised ones, calc_g_loss :199).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from . import autograd as A
from . import kernels as K
from . import nets
from .autograd import Ctx, ParamGroup

Tensor = torch.Tensor

DEFAULT_LOSS = dict(distortion=50.0, perceptual=1.0, gan=0.01, code_distortion=1.0, code_ce=0.5)


def allreduce_mean_(flat: Tensor, dist, bucket_bytes: int = 64 << 20) -> int:
    """Average a flat gradient buffer over the ranks with bucketed all-reduces (async, joined at the end).  Ring collectives
    over xGMI are per-link bound, so a few large buckets beat many small ones; 64 MiB keeps three or so in flight for the
    134 MB generator gradient.  Returns the number of buckets."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    n = flat.numel()
    step = max(1, bucket_bytes // 4)
    works = []
    for off in range(0, n, step):
        works.append(dist.all_reduce(flat[off:off + step], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    if flat.is_cuda:
        K.ew(11, None, flat, w=1.0 / dist.get_world_size(), out=flat)
    else:
        flat.mul_(1.0 / dist.get_world_size())        # CPU (gloo tests): plumbing only
    return len(works)


def loss_anomaly(total: float, dist, device=None) -> bool:
    """base_trainer.py:235-245 skips a step whose total loss is non-finite or > 1e4.  With several ranks the decision must be
    COLLECTIVE: a rank that returned early while the others entered the gradient all-reduce would leave them blocked in RCCL
    forever (and let the Adam / scheduler step counts diverge).  One 1-element MAX all-reduce of the local flag: every rank
    skips or none does."""
    bad = (not math.isfinite(total)) or total > 1e4
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bad
    flag = torch.tensor([1.0 if bad else 0.0], dtype=torch.float32, device=device if device is not None else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(flag.item() > 0)


class MultiStepLR:
    def __init__(self, base_lr: float, milestones: List[int], gamma: float):
        self.base_lr, self.milestones, self.gamma, self.last_epoch = base_lr, sorted(milestones), gamma, 0

    def lr(self) -> float:
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.last_epoch)

    def step(self) -> None:
        self.last_epoch += 1


class Adam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) on a ParamGroup's flat buffers."""

    def __init__(self, group: ParamGroup, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.group, self.lr, self.betas, self.eps, self.t = group, lr, betas, eps, 0

    def step(self, lr: Optional[float] = None, gscale: Optional[Tensor] = None) -> None:
        self.t += 1
        g = self.group
        K.adam_step(g.flat, g.grad, g.m, g.v, self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps, self.t, gscale)
        g.refresh_plans()


class DualBetaCondGanDistortionVqCodeTrainer:
    def __init__(self, model, discriminator, lr_g: float = 1e-4, lr_d: float = 1e-4, milestones=(300000,), gamma: float = 0.1,
                 clip_max_norm: Optional[float] = 1.0, loss_weights: Optional[Dict[str, float]] = None, sample_beta_batch: bool = True,
                 dist=None, seed: int = 0, d_milestones=None, d_gamma: Optional[float] = None, lpips_state: Optional[Dict[str, Tensor]] = None):
        self.model, self.D = model, discriminator
        dev = next(model.decoder.parameters()).device
        self.device = dev
        # only decoder, vq_estimator, fusion_module are trained (:47-52)
        self.g_group = ParamGroup([model.decoder, model.vq_estimator, model.fusion_module], dev)
        self.d_group = ParamGroup([discriminator], dev)
        self.g_opt, self.d_opt = Adam(self.g_group, lr_g), Adam(self.d_group, lr_d)
        # config/exp1_stage1_3.yaml:43-60: g_scheduler and d_scheduler are separate YAML entries (same values in the shipped files)
        self.g_sched = MultiStepLR(lr_g, list(milestones), gamma)
        self.d_sched = MultiStepLR(lr_d, list(milestones if d_milestones is None else d_milestones), gamma if d_gamma is None else d_gamma)
        self.clip = clip_max_norm
        self.w = dict(DEFAULT_LOSS)
        if loss_weights:
            self.w.update(loss_weights)
        self.sample_beta_batch = sample_beta_batch
        self.dist = dist
        self.rng = np.random.RandomState(seed)
        self.last_fake: Optional[Tensor] = None
        self.lpips = None
        if self.w["perceptual"] > 0:
            from .lpips import LPIPSAlex
            self.lpips = LPIPSAlex(seed=0).to(dev)
            self.lpips_is_synthetic = lpips_state is None
            if lpips_state is not None:
                self.load_lpips_state(lpips_state)

    def load_lpips_state(self, sd: Dict[str, Tensor]) -> None:
        """Load a state dict of the `lpips` package (LPIPS(net='alex').state_dict(): net.sliceK.I.weight/bias, linK.model.1.weight,
        scaling_layer.shift/scale) into the perceptual network, strictly by key and shape."""
        own = self.lpips.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"lpips state dict lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for k, v in own.items():
            if tuple(sd[k].shape) != tuple(v.shape):
                raise ValueError(f"lpips state dict: {k} has shape {tuple(sd[k].shape)}, expected {tuple(v.shape)}")
            v.copy_(sd[k].to(v.device, dtype=v.dtype))
        for m in self.lpips.modules():
            if hasattr(m, "_plan"):
                m._plan = None
        self.lpips_is_synthetic = False

    # ---------------------------------------------------------------- training state (base_trainer.py:178-214 saves optimizer + scheduler state)
    def training_state(self, current_iter: int) -> Dict:
        st = {"iter": int(current_iter), "rng": self.rng.get_state()}
        for tag, grp, opt_, sch in (("g", self.g_group, self.g_opt, self.g_sched), ("d", self.d_group, self.d_opt, self.d_sched)):
            st[tag] = {"m": grp.m.detach().cpu().clone(), "v": grp.v.detach().cpu().clone(), "t": int(opt_.t), "last_epoch": int(sch.last_epoch),
                       "numel": int(grp.numel())}
        return st

    def load_training_state(self, st: Dict) -> int:
        for tag, grp, opt_, sch in (("g", self.g_group, self.g_opt, self.g_sched), ("d", self.d_group, self.d_opt, self.d_sched)):
            s_ = st[tag]
            if int(s_["numel"]) != grp.numel():
                raise ValueError(f"training state: {tag} group has {s_['numel']} parameters, this model {grp.numel()}")
            grp.m.copy_(s_["m"].to(grp.m.device)); grp.v.copy_(s_["v"].to(grp.v.device))
            opt_.t, sch.last_epoch = int(s_["t"]), int(s_["last_epoch"])
        if "rng" in st:
            self.rng.set_state(st["rng"])
        return int(st["iter"])

    def _drop_inference_graphs(self) -> None:
        """The generator's weights moved in place: hipGraphs captured by compress_batch / decompress_batch before that replay the old
        packed weights -- forget them (the next inference call re-captures)."""
        g = getattr(self.model, "_graphs", None)
        if g is not None:
            g.clear()

    def resync_parameters(self) -> None:
        """After a state dict was loaded into the modules: parameters are views of the flat buffers, so load_state_dict's copy_
        already wrote them; cached packed weights are stale."""
        self.g_group.refresh_plans(); self.d_group.refresh_plans()
        self._drop_inference_graphs()

    # hyperprior_dc_vic_model.py:99-110
    def sample_selected_beta_pair(self, n: int) -> Tuple[Tensor, Tensor]:
        m = self.model
        i = self.rng.randint(0, len(m.selected_beta_rate), n)
        return (torch.Tensor([m.selected_beta_rate[k] for k in i]).float(), torch.Tensor([m.selected_beta_vq[k] for k in i]).float())

    @torch.no_grad()
    def generator_forward(self, ctx: Ctx, real: Tensor, vq_indices: Optional[Tensor], beta_rate, beta_vq):
        """run_comp_model (:116-133) -> forward (hyperprior_dc_vic_model.py:208-274, is_train, fix_entropy_models)."""
        m = self.model
        x = real.to(self.device, dtype=torch.float32).contiguous()
        gt_lat, gt_idx, feat = m.vq_encode(x, vq_indices, want_feat=True)                      # frozen VQGAN, no grad
        y = m.comp_encode(x, gt_lat, gt_idx, enc_kwargs=dict(beta_1=beta_rate, beta_2=beta_vq), feat=feat)
        e = m._entropy_encode_side(y, want_symbols=False)                                      # fix_entropy_models: no grad
        y_hat = e["y_hat"]
        feat_1, feats = nets.decoder_get_feats(ctx, m.decoder, y_hat, beta_rate, beta_vq)
        pred_embed, logits = nets.estimator_forward(ctx, m.vq_estimator, feat_1)
        pq = m.vq_model.post_quant_conv
        out_idx, lat = ops.argmax_lut(logits.data, m.vq_model.quantize.embedding.weight, pq.weight.reshape(pq.out_channels, -1).contiguous(), pq.bias)
        fake = nets.fusion_decode(ctx, m.fusion_module, m.vq_model.decoder, A.const(lat), feats, w=1.0)
        N, _, H, W = x.shape
        qbpp = float((e["bits_y"].double().sum() + e["bits_z"].double().sum()).item()) / (N * H * W)
        return dict(real=x, fake=fake, pred_embed=pred_embed, logits=logits, gt_vq_latent=gt_lat, gt_vq_indices=gt_idx, out_vq_indices=out_idx,
                    y_hat=y_hat, qbpp=qbpp)

    def calc_g_loss(self, ctx: Ctx, o: Dict, beta_rate, beta_vq) -> Dict[str, Tensor]:
        """:192-234; each term seeds its gradient on the tape."""
        w = self.w
        log = {}
        log["distortion"] = A.mse_loss(ctx, o["fake"], o["real"], w["distortion"] * 0.25)        # MSELoss on [0,1]: ((a+1)/2-(b+1)/2)^2
        if self.lpips is not None:
            from .lpips import lpips_loss
            log["perceptual"] = lpips_loss(ctx, self.lpips, o["real"], o["fake"], w["perceptual"])
        else:
            log["perceptual"] = torch.zeros(1, device=self.device)
        g_fake = nets.discriminator_forward(ctx, self.D, o["fake"], beta_rate, beta_vq)
        log["adv"] = A.bce_logits_loss(ctx, g_fake, True, w["gan"])
        log["code_distortion"] = A.mse_loss(ctx, o["pred_embed"], o["gt_vq_latent"], w["code_distortion"])
        log["code_ce"] = A.cross_entropy_loss(ctx, o["logits"], o["gt_vq_indices"], w["code_ce"])
        return log

    def optimize_parameters(self, current_iter: int, data_dict: Dict) -> Optional[Dict[str, float]]:
        real = data_dict["real_images"]
        vq_indices = data_dict.get("vq_indices")
        n = real.shape[0]
        beta_rate, beta_vq = data_dict.get("beta_rate"), data_dict.get("beta_vq")
        if beta_rate is None or beta_vq is None:
            beta_rate, beta_vq = self.sample_selected_beta_pair(n if self.sample_beta_batch else 1)
        # ---------------------------------------------------------------- train G
        self.g_group.zero_grad()
        ctx = Ctx([self.g_group])                      # D's parameters receive no gradient here (requires_grad_(False), :146)
        o = self.generator_forward(ctx, real, vq_indices, beta_rate, beta_vq)
        g_log = self.calc_g_loss(ctx, o, beta_rate, beta_vq)
        total = sum(float(v.item()) for v in g_log.values())
        if loss_anomaly(total, self.dist, self.device):  # base_trainer.py:235-245: skip the step -- on EVERY rank or on none
            ctx.tape = []
            return None
        ctx.backward()
        allreduce_mean_(self.g_group.grad, self.dist)
        gscale = None
        if self.clip:
            gscale = K.clip_scale(K.reduce_loss(2, self.g_group.grad, None, 1.0), self.clip)
        self.g_opt.step(self.g_sched.lr(), gscale)
        self._drop_inference_graphs()
        lr_now = self.g_sched.lr()
        self.g_sched.step()
        # ---------------------------------------------------------------- train D
        self.d_group.zero_grad()
        dctx = Ctx([self.d_group])
        fake_det = o["fake"].data                      # .detach()
        self.last_fake = fake_det
        d_real = nets.discriminator_forward(dctx, self.D, A.const(o["real"]), beta_rate, beta_vq)
        d_fake = nets.discriminator_forward(dctx, self.D, A.const(fake_det), beta_rate, beta_vq)
        l_real = A.bce_logits_loss(dctx, d_real, True, 0.5)
        l_fake = A.bce_logits_loss(dctx, d_fake, False, 0.5)
        dctx.backward()
        allreduce_mean_(self.d_group.grad, self.dist)
        self.d_opt.step(self.d_sched.lr())
        self.d_sched.step()
        vq_acc = float((o["out_vq_indices"] == o["gt_vq_indices"]).float().mean().item())
        log = {k: float(v.item()) for k, v in g_log.items()}
        log.update(total=total, qbpp=o["qbpp"], vq_acc=vq_acc, lr=lr_now, d_real=float(l_real.item()), d_fake=float(l_fake.item()),
                   d_total=float(l_real.item()) + float(l_fake.item()),
                   out_d_real=float(d_real.data.mean().item()), out_d_fake=float(d_fake.data.mean().item()))
        return log
