"""A small reverse-mode tape over the HIP kernels -- what torch.autograd does for the reference's training step
(src/trainer/dual_cond_gan_distortion_vq_code_trainer.py:135-190: `l_total.backward()`, `d_loss.backward()`).

`Var` wraps a device tensor and its gradient; every differentiable op runs the inference kernel forward, records a
closure on the tape, and the closure launches the backward kernels (csrc/train.hip) or -- for the data gradient of a
convolution -- the forward convolution kernel on transposed / flipped weights.  Parameters live in flat buffers
(`ParamGroup`): weight gradients are accumulated straight into the flat gradient buffer, so Adam, gradient clipping and
the data-parallel all-reduce each see ONE contiguous tensor.  No torch autograd, no torch compute op on the path.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..layers import Conv2d, ConvTranspose2d, GroupNorm, LayerNormC, Linear
from . import kernels as K

Tensor = torch.Tensor


class Var:
    __slots__ = ("data", "grad", "needs_grad")

    def __init__(self, data: Tensor, needs_grad: bool = True):
        self.data = data
        self.grad: Optional[Tensor] = None
        self.needs_grad = needs_grad

    @property
    def shape(self):
        return self.data.shape


def const(t: Tensor) -> Var:
    return Var(t, needs_grad=False)


class ParamGroup:
    """The trainable parameters of some modules as views of ONE flat fp32 buffer (+ flat grad / Adam moments)."""

    def __init__(self, modules: Sequence[nn.Module], device):
        self.params: List[nn.Parameter] = []
        seen = set()
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen and p.dtype == torch.float32:
                    seen.add(id(p))
                    self.params.append(p)
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        self._gview: Dict[int, Tensor] = {}
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)           # the module now reads / Adam writes the flat buffer
            self._gview[id(p)] = self.grad[off:off + k].view(p.shape)
            off += k
        self.modules = list(modules)
        self.step_count = 0

    def numel(self) -> int:
        return self.flat.numel()

    def grad_of(self, p: nn.Parameter) -> Optional[Tensor]:
        return self._gview.get(id(p))

    def zero_grad(self) -> None:
        self.grad.zero_()

    def refresh_plans(self) -> None:
        """The flat buffer was updated in place (Adam kernel): cached packed weights / beta vectors are stale."""
        for mod in self.modules:
            for m in mod.modules():
                if hasattr(m, "_graphs"):
                    m._graphs.clear()          # (a whole model in the group: its captured hipGraphs hold the old packed weights)
                if hasattr(m, "_plan"):
                    m._plan = None
                if hasattr(m, "_qkv_plan"):
                    m._qkv_plan = None
                if getattr(m, "_dgrad", None) is not None:
                    # data-gradient plan cached while this module was NOT trainable in some Ctx (the discriminator inside the
                    # generator step): it holds a flipped / transposed COPY of the weights, stale once Adam moved them
                    m._dgrad = None
                if hasattr(m, "invalidate_caches"):
                    m.invalidate_caches()


class Ctx:
    """One forward/backward pass: the tape and the parameter groups that receive weight gradients."""

    def __init__(self, groups: Sequence[ParamGroup] = ()):
        self.tape: List[Callable[[], None]] = []
        self.groups = list(groups)

    def pgrad(self, p: Optional[nn.Parameter]) -> Optional[Tensor]:
        if p is None:
            return None
        for g in self.groups:
            v = g.grad_of(p)
            if v is not None:
                return v
        return None

    def backward(self) -> None:
        for fn in reversed(self.tape):
            fn()
        self.tape = []


def acc(v: Var, g: Tensor) -> None:
    """v.grad += g (out of place: a gradient tensor may be shared by several consumers)."""
    if not v.needs_grad:
        return
    if v.grad is None:
        v.grad = g
    else:
        v.grad = K.ew(10, None, v.grad.contiguous(), g.contiguous())


def _dense(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else ops.copy_planes(torch.empty(t.shape, dtype=torch.float32, device=t.device), t, t.shape[2], t.shape[3])


# ------------------------------------------------------------------------------------------- convolution family
def _dgrad_plan(mod) -> ops.ConvPlan:
    w = mod.weight.detach()
    if isinstance(mod, Linear):
        return ops.ConvPlan(w.t().contiguous(), None, "conv")
    if isinstance(mod, ConvTranspose2d):
        if mod.kernel_size != 5:
            raise NotImplementedError("only the k5/s2 transposed convolution is trained")
        # adjoint of ConvTranspose2d(k5, s2, p2, op1) = Conv2d(k5, s2, p2) with the same [Cin, Cout, k, k] tensor read as [out, in, k, k]
        return ops.ConvPlan(w, None, "conv", stride=2, pad=(2, 2))
    k, s, p = mod.kernel_size, mod.stride, mod.padding
    if mod.asym_pad:
        raise NotImplementedError("the ldm Downsample conv is encoder-side only (never differentiated)")
    if s == 1:
        plan = ops.ConvPlan(w.flip(2, 3).transpose(0, 1).contiguous(), None, "conv", pad=(k - 1 - p, k - 1 - p))
        # the data gradient of a Conv2d(k3, s1, p1) is again one: Winograd wherever the forward layer opted in (layers.allow_winograd)
        plan.wino = bool(getattr(mod, "wino", False)) and (k, p) == (3, 1)
        plan.wino44 = plan.wino and bool(getattr(mod, "wino44", False))   # (F(4x4) where the forward layer may use it: decoder + fusion)
        return plan
    if (k, s, p) == (4, 2, 1):
        # adjoint = ConvTranspose2d(k4, s2, p1): output row 2m+py takes (input row, ky) in {(m-1, 3), (m, 1)} for py = 0 and
        # {(m, 2), (m+1, 0)} for py = 1 (oy = 2 iy - 1 + ky); four 2x2 sub-pixel convolutions
        ks = ((3, 1), (2, 0))
        wt = w.transpose(0, 1)                                   # [Cin_conv (out of the adjoint), Cout_conv, ky, kx]
        ph = []
        for py in (0, 1):
            for px in (0, 1):
                ph.append(wt[:, :, list(ks[py])][:, :, :, list(ks[px])].contiguous())
        return ops.ConvPlan.from_phases2(ph)
    raise NotImplementedError(f"conv dgrad for k{k} s{s} p{p}")


def conv(ctx: Ctx, x: Var, mod, act: int = ops.ACT_NONE) -> Var:
    """Conv2d / ConvTranspose2d / Linear stand-ins of dc_vic_amd.layers; bias and (optionally) ReLU / LeakyReLU fused forward."""
    if act not in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LRELU02):
        raise ValueError("only sign-derivable activations may be fused into a differentiable conv")
    xd = x.data
    gw = ctx.pgrad(mod.weight)
    trainable = gw is not None
    ups = isinstance(mod, Conv2d) and mod.upsample
    if ups:
        # differentiable path: explicit nearest x2 then the plain 3x3 convolution (ldm model.py:53-57)
        xin = K.resample2(_dense(xd), down=False)
        cached = None if trainable else getattr(mod, "_plain_plan", None)
        plan = cached[1] if cached is not None and cached[0] == mod._key() else None
        if plan is None:
            plan = ops.ConvPlan(mod.weight, mod.bias, "conv", pad=(1, 1))
            plan.wino = bool(getattr(mod, "wino", False))
            plan.wino44 = plan.wino and bool(getattr(mod, "wino44", False))
            if not trainable:
                mod._plain_plan = (mod._key(), plan)   # frozen weights (the VQGAN decoder's Upsample convs): pack once per weight version
    else:
        xin = xd
        if trainable:        # the weights move every step: never reuse a cached pack
            if isinstance(mod, ConvTranspose2d):
                plan = ops.ConvPlan(mod.weight, mod.bias, "convT")
            elif isinstance(mod, Linear):
                plan = ops.ConvPlan(mod.weight, mod.bias, "conv")
            else:
                plan = ops.ConvPlan(mod.weight, mod.bias, "conv", stride=mod.stride, pad=(mod.padding, mod.padding))
                plan.wino = bool(getattr(mod, "wino", False))    # (the SFT fusion blocks train; their 3x3 convs stay on Winograd)
                plan.wino44 = plan.wino and bool(getattr(mod, "wino44", False))
        else:
            plan = mod._get_plan()
    y = plan(xin, act=act)
    out = Var(y)

    def back():
        g = out.grad
        if g is None:
            return
        g = _dense(g)
        if act != ops.ACT_NONE:
            g = K.ew(0, g, y, act=act)
        if x.needs_grad:
            cached = getattr(mod, "_dgrad", None)
            dp = cached[1] if (not trainable and cached is not None and cached[0] == mod._key()) else None
            if dp is None:
                dp = _dgrad_plan(mod)
                if not trainable:
                    mod._dgrad = (mod._key(), dp)          # keyed by the weights' version: a reloaded checkpoint rebuilds it
            dx = dp(g)
            if ups:
                dx = K.resample2(dx, down=True)
            acc(x, dx)
        if trainable:
            xs = _dense(xin)
            if isinstance(mod, ConvTranspose2d):
                K.conv_wgrad(xs, g, gw, 5, 5, 2, 2)                 # roles swapped: result is [Cin][Cout][k][k]
            elif isinstance(mod, Linear):
                K.conv_wgrad(g, xs, gw, 1, 1, 1, 0)
            else:
                K.conv_wgrad(g, xs, gw, mod.kernel_size, mod.kernel_size, mod.stride, mod.padding)
            gb = ctx.pgrad(mod.bias)
            if gb is not None:
                K.sum_rows(K.chan_reduce(g), gb, accumulate=True)
    ctx.tape.append(back)
    return out


# ------------------------------------------------------------------------------------------- normalisation
def group_norm(ctx: Ctx, x: Var, mod: GroupNorm, act: int = ops.ACT_NONE) -> Var:
    xd = x.data
    y = ops.groupnorm(xd, mod.weight, mod.bias, mod.num_groups, mod.eps, act)
    out = Var(y)

    def back():
        if out.grad is None:
            return
        dx, dg, db = K.groupnorm_bwd(xd, out.grad, mod.weight, mod.bias, mod.num_groups, mod.eps, act)
        acc(x, dx)
        gw, gb = ctx.pgrad(mod.weight), ctx.pgrad(mod.bias)
        if gw is not None:
            K.sum_rows(dg, gw, accumulate=True)
            K.sum_rows(db, gb, accumulate=True)
    ctx.tape.append(back)
    return out


def layer_norm_c(ctx: Ctx, x: Var, mod: LayerNormC) -> Var:
    xd = _dense(x.data)
    y = ops.layernorm_c(xd, mod.weight, mod.bias, mod.eps)
    out = Var(y)

    def back():
        if out.grad is None:
            return
        dx, part = K.layernorm_c_bwd(xd, _dense(out.grad), mod.weight, mod.eps)
        acc(x, dx)
        gw, gb = ctx.pgrad(mod.weight), ctx.pgrad(mod.bias)
        if gw is not None:
            Cc = xd.shape[1]
            tmp = torch.empty(2 * Cc, dtype=torch.float32, device=xd.device)
            K.sum_rows(part, tmp, accumulate=False)
            K.ew(10, None, gw, tmp[:Cc], out=gw)
            K.ew(10, None, gb, tmp[Cc:], out=gb)
    ctx.tape.append(back)
    return out


# ------------------------------------------------------------------------------------------- elementwise
def activation(ctx: Ctx, x: Var, act: int) -> Var:
    xd = _dense(x.data)
    y = ops.activation(xd, act)
    out = Var(y)
    ref = xd if act in (ops.ACT_SWISH, ops.ACT_GELU) else y

    def back():
        if out.grad is not None:
            acc(x, K.ew(0, _dense(out.grad), ref, act=act))
    ctx.tape.append(back)
    return out


def add(ctx: Ctx, a: Var, b: Var) -> Var:
    out = Var(ops.add(_dense(a.data), _dense(b.data)))

    def back():
        if out.grad is not None:
            acc(a, out.grad)
            acc(b, out.grad)
    ctx.tape.append(back)
    return out


def nlam_gate(ctx: Ctx, x: Var, t: Var, a: Var) -> Var:
    """x + t * sigmoid(a)   (ChengNLAM, cheng_nlam.py:23-27)."""
    xd, td, ad = _dense(x.data), _dense(t.data), _dense(a.data)
    out = Var(ops.add_mul_sigmoid(xd, td, ad))

    def back():
        g = out.grad
        if g is None:
            return
        g = _dense(g)
        acc(x, g)
        acc(t, K.ew(2, g, ad))
        acc(a, K.ew(3, g, td, ad))
    ctx.tape.append(back)
    return out


def sft(ctx: Ctx, dec: Var, scale: Var, shift: Var, w: float = 1.0) -> Var:
    """dec + w * (dec * scale + shift)   (FuseSftBlock, codeformer_layers.py:65-66)."""
    dd, sd, hd = _dense(dec.data), _dense(scale.data), _dense(shift.data)
    out = Var(ops.sft(dd, sd, hd, w=w))

    def back():
        g = out.grad
        if g is None:
            return
        g = _dense(g)
        acc(dec, K.ew(4, g, sd, w=w))
        acc(scale, K.ew(5, g, dd, w=w))
        acc(shift, K.ew(6, g, w=w))
    ctx.tape.append(back)
    return out


def chan_affine(ctx: Ctx, x: Var, s: Var, t: Var, add_x: bool = False) -> Var:
    """x * (1 + s[n][c]) + t[n][c] (+ x)   (BetaScaleShiftModule.forward; s, t: [B, C, 1, 1] with B in {1, N})."""
    xd = _dense(x.data)
    N, Cc = xd.shape[:2]
    sv, tv = s.data.reshape(s.data.shape[0], Cc).contiguous(), t.data.reshape(t.data.shape[0], Cc).contiguous()
    out = Var(ops.chan_affine(xd, sv, tv, add_=xd if add_x else None))

    def back():
        g = out.grad
        if g is None:
            return
        g = _dense(g)
        if x.needs_grad:
            dx = K.ew(7, g, sv, vec_bs=Cc if sv.shape[0] > 1 else 0)
            if add_x:
                dx = K.ew(10, None, dx, g)
            acc(x, dx)
        ds, dt = K.chan_reduce(g, xd), K.chan_reduce(g)                   # [N, C]
        if sv.shape[0] == 1:
            ds1, dt1 = torch.empty((1, Cc), dtype=torch.float32, device=g.device), torch.empty((1, Cc), dtype=torch.float32, device=g.device)
            K.sum_rows(ds, ds1.view(-1), accumulate=False)
            K.sum_rows(dt, dt1.view(-1), accumulate=False)
            ds, dt = ds1, dt1
        acc(s, ds.view(s.data.shape))
        acc(t, dt.view(t.data.shape))
    ctx.tape.append(back)
    return out


def upsample2(ctx: Ctx, x: Var) -> Var:
    out = Var(K.resample2(_dense(x.data), down=False))

    def back():
        if out.grad is not None:
            acc(x, K.resample2(_dense(out.grad), down=True))
    ctx.tape.append(back)
    return out


def cat_channels(ctx: Ctx, parts: Sequence[Var]) -> Var:
    N, _, H, W = parts[0].shape
    Ct = sum(p.shape[1] for p in parts)
    buf = torch.empty((N, Ct, H, W), dtype=torch.float32, device=parts[0].data.device)
    off = 0
    offs = []
    for p in parts:
        c = p.shape[1]
        ops.copy_planes(buf[:, off:off + c], p.data, H, W)
        offs.append((off, c))
        off += c
    out = Var(buf)

    def back():
        if out.grad is None:
            return
        for p, (o, c) in zip(parts, offs):
            if p.needs_grad:
                acc(p, _dense(out.grad[:, o:o + c]))
    ctx.tape.append(back)
    return out


# ------------------------------------------------------------------------------------------- attention
def attn_single_head(ctx: Ctx, qkv: Var, Cc: int) -> Var:
    """ldm AttnBlock core (model.py:186-196): forward = the fused flash kernel; backward recomputes the probabilities
    (bgemm + column softmax) and forms dV, dP, dS, dQ, dK with the strided batched GEMM."""
    qd = _dense(qkv.data)
    N, _, H, W = qd.shape
    HW = H * W
    out = Var(ops.attn_fused(qd, Cc))
    scale = float(int(Cc) ** (-0.5))

    def back():
        g = out.grad
        if g is None:
            return
        g = _dense(g)                                              # dO [N, C, HW]
        bs = 3 * Cc * HW
        q, k, v = qd[:, :Cc], qd[:, Cc:2 * Cc], qd[:, 2 * Cc:]
        St = torch.empty((N, HW, HW), dtype=torch.float32, device=qd.device)       # St[j][i] = P(query i -> key j)
        ops.bgemm(k, (bs, 1, HW), q, (bs, HW, 1), St, (HW * HW, HW), N, HW, HW, Cc, alpha=scale)
        ops.softmax_c_(St, N, HW, HW)
        dqkv = torch.empty_like(qd)
        dq, dk, dv = dqkv[:, :Cc], dqkv[:, Cc:2 * Cc], dqkv[:, 2 * Cc:]
        # dV[c][j] = sum_i dO[c][i] St[j][i]
        ops.bgemm(g, (Cc * HW, HW, 1), St, (HW * HW, 1, HW), dv, (bs, HW), N, Cc, HW, HW)
        # dSt[j][i] = sum_c V[c][j] dO[c][i]
        dSt = torch.empty_like(St)
        ops.bgemm(v, (bs, 1, HW), g, (Cc * HW, HW, 1), dSt, (HW * HW, HW), N, HW, HW, Cc)
        dS = K.softmax_c_bwd(St, dSt, scale)                        # includes the c^-0.5 factor
        # dQ[c][i] = sum_j K[c][j] dS[j][i] ; dK[c][j] = sum_i Q[c][i] dS[j][i]
        ops.bgemm(k, (bs, HW, 1), dS, (HW * HW, HW, 1), dq, (bs, HW), N, Cc, HW, HW)
        ops.bgemm(q, (bs, HW, 1), dS, (HW * HW, 1, HW), dk, (bs, HW), N, Cc, HW, HW)
        acc(qkv, dqkv)
    ctx.tape.append(back)
    return out


def swin_attention(ctx: Ctx, qkv: Var, table: nn.Parameter, heads: int, ws: int, shift: int) -> Var:
    qd = _dense(qkv.data)
    out = Var(ops.swin_attn(qd, table, heads, ws, shift))

    def back():
        if out.grad is None:
            return
        gt = ctx.pgrad(table)
        dt = gt if gt is not None else torch.zeros_like(table)
        acc(qkv, K.swin_attn_bwd(qd, _dense(out.grad), table, dt, heads, ws, shift, accumulate=gt is not None))
    ctx.tape.append(back)
    return out


# ------------------------------------------------------------------------------------------- losses (seed the gradients)
def mse_loss(ctx: Ctx, a: Var, target: Tensor, weight: float) -> Tensor:
    """weight * mean((a - target)^2): value as a 1-element device tensor, gradient seeded into `a`."""
    ad, td = _dense(a.data), _dense(target)
    n = ad.numel()
    val = K.reduce_loss(0, ad, td, weight / n)
    acc(a, K.ew(8, None, ad, td, w=weight / n))
    return val


def bce_logits_loss(ctx: Ctx, x: Var, is_real: bool, weight: float) -> Tensor:
    """weight * BCEWithLogits(x, 1 or 0), mean reduction (src/losses/gan_loss.py:10-32)."""
    xd = _dense(x.data)
    n = xd.numel()
    val = K.reduce_loss(1, xd, None, weight / n, target=1 if is_real else 0)
    acc(x, K.ew(9, None, xd, w=weight / n, act=1 if is_real else 0))
    return val


def cross_entropy_loss(ctx: Ctx, logits: Var, target: Tensor, weight: float) -> Tensor:
    """weight * CrossEntropy(logits [N, C, H, W], target [N, H, W]), mean over N*H*W (cross_entropy_loss.py:10-28)."""
    ld = _dense(logits.data)
    N, Cc, H, W = ld.shape
    m = N * H * W
    nll, dl = K.cross_entropy(ld, target.contiguous(), weight / m, want_grad=True)
    val = K.reduce_loss(3, nll, None, weight / m)
    acc(logits, dl)
    return val
