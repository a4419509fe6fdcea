"""LPIPS(net='alex') perceptual loss on HIP kernels -- the `perceptual_loss` of config/exp1_stage1_3.yaml:66-69
(src/losses/perceptual_loss.py:10-30 wraps `lpips.LPIPS`).

PARITY UNPINNED: neither the `lpips` package nor its AlexNet / linear-head weights are in the reference tree and they cannot
be fetched.  The architecture is restated from the published model (AlexNet features, taps after each ReLU; input scaling
layer; unit-normalise over channels; squared difference; non-negative 1x1 heads; spatial mean; sum of the 5 taps) with the
package's state-dict key names, so a real `lpips` state dict loads by key; without one the weights are deterministic synthetic
ones.  The 11x11 / stride-4 / pad-2 stem runs as a 3x3 / stride-1 convolution over the 48 channels of a zero-padded
space-to-depth(4) image (kernel zero-padded to 12x12) -- same sums, conv kernels limited to 25 taps."""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.nn as nn

from .. import ops
from .._lib import check, lib
from ..layers import Conv2d
from ..ops import _chk4, _p, _stream
from . import autograd as A
from . import kernels as K
from .autograd import Ctx, Var

Tensor = torch.Tensor
CHNS = (64, 192, 384, 256, 256)


class _Slice(nn.Sequential):
    pass


class LPIPSAlex(nn.Module):
    def __init__(self, seed: int = 0):
        super().__init__()
        net = nn.Module()
        # torchvision alexnet.features indices inside each lpips slice: slice1 = [conv0, relu], slice2 = [pool, conv3, relu], ...
        net.slice1 = _Slice(); net.slice1.add_module("0", Conv2d(3, 64, 11, 4, 2))
        net.slice2 = _Slice(); net.slice2.add_module("3", Conv2d(64, 192, 5, 1, 2))
        net.slice3 = _Slice(); net.slice3.add_module("6", Conv2d(192, 384, 3, 1, 1))
        net.slice4 = _Slice(); net.slice4.add_module("8", Conv2d(384, 256, 3, 1, 1))
        net.slice5 = _Slice(); net.slice5.add_module("10", Conv2d(256, 256, 3, 1, 1))
        self.net = net
        for i, c in enumerate(CHNS):
            lin = nn.Module()
            lin.model = nn.Sequential(nn.Identity(), Conv2d(c, 1, 1, bias=False))
            setattr(self, f"lin{i}", lin)
        sl = nn.Module()
        sl.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        sl.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])
        self.scaling_layer = sl
        g = torch.Generator().manual_seed(1000 + seed)
        for m in self.convs():
            fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            m.weight.data.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
        for i, c in enumerate(CHNS):
            w = getattr(self, f"lin{i}").model[1].weight
            w.data.copy_(torch.randn(w.shape, generator=g).abs() / c)
        self._stem: List[Conv2d] = []            # derived 48 -> 64 3x3 stem (not a registered sub-module)

    def convs(self):
        n = self.net
        return [n.slice1[0], n.slice2[0], n.slice3[0], n.slice4[0], n.slice5[0]]

    def stem(self) -> Conv2d:
        w11 = self.net.slice1[0].weight
        if not self._stem or self._stem[0].weight.device != w11.device or self._stem_key != (w11.data_ptr(), w11._version):
            w12 = torch.zeros((64, 3, 12, 12), dtype=torch.float32, device=w11.device)
            w12[:, :, :11, :11] = w11.detach()
            # ky = 4 a + dy -> [o, c, a, dy, b, dx] -> [o, (c, dy, dx), a, b]
            w = w12.view(64, 3, 3, 4, 3, 4).permute(0, 1, 3, 5, 2, 4).reshape(64, 48, 3, 3).contiguous()
            m = Conv2d(48, 64, 3, 1, 0).to(w11.device)
            m.weight.data.copy_(w); m.bias.data.copy_(self.net.slice1[0].bias.detach())
            self._stem = [m]
            self._stem_key = (w11.data_ptr(), w11._version)
        return self._stem[0]


def s2d(ctx: Ctx, x: Var, r: int, pad: int) -> Var:
    xd = A._dense(x.data)
    N, Cc, H, W = _chk4(xd, "s2d x")
    Ho, Wo = (H + 2 * pad) // r, (W + 2 * pad) // r
    y = torch.empty((N, Cc * r * r, Ho, Wo), dtype=torch.float32, device=xd.device)
    check(lib().dcvic_s2d_f32(_p(xd), _p(y), C.c_longlong(N * Cc), H, W, r, pad, 0, _stream()), "s2d")
    out = Var(y)

    def back():
        if out.grad is None or not x.needs_grad:
            return
        dx = torch.empty_like(xd)
        check(lib().dcvic_s2d_f32(_p(A._dense(out.grad)), _p(dx), C.c_longlong(N * Cc), H, W, r, pad, 1, _stream()), "s2d_bwd")
        A.acc(x, dx)
    ctx.tape.append(back)
    return out


def maxpool3s2(ctx: Ctx, x: Var) -> Var:
    xd = A._dense(x.data)
    N, Cc, H, W = _chk4(xd, "maxpool x")
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    y = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=xd.device)
    am = torch.empty((N, Cc, Ho, Wo), dtype=torch.uint8, device=xd.device)
    check(lib().dcvic_maxpool3s2_f32(_p(xd), _p(y), _p(am), None, None, C.c_longlong(N * Cc), H, W, _stream()), "maxpool")
    out = Var(y)

    def back():
        if out.grad is None or not x.needs_grad:
            return
        dx = torch.empty_like(xd)
        check(lib().dcvic_maxpool3s2_f32(None, None, _p(am), _p(A._dense(out.grad)), _p(dx), C.c_longlong(N * Cc), H, W, _stream()), "maxpool_bwd")
        A.acc(x, dx)
    ctx.tape.append(back)
    return out


def features(ctx: Ctx, L: LPIPSAlex, img: Var) -> List[Var]:
    """scaling layer + the five AlexNet taps."""
    sh, sc = L.scaling_layer.shift, L.scaling_layer.scale
    s = A.const((1.0 / sc - 1.0).contiguous())                    # x * (1 + s) + t = (x - shift) / scale
    t = A.const((-sh / sc).contiguous())
    x = A.chan_affine(ctx, img, s, t)
    f1 = A.conv(ctx, s2d(ctx, x, 4, 2), L.stem(), act=ops.ACT_RELU)
    f2 = A.conv(ctx, maxpool3s2(ctx, f1), L.net.slice2[0], act=ops.ACT_RELU)
    f3 = A.conv(ctx, maxpool3s2(ctx, f2), L.net.slice3[0], act=ops.ACT_RELU)
    f4 = A.conv(ctx, f3, L.net.slice4[0], act=ops.ACT_RELU)
    f5 = A.conv(ctx, f4, L.net.slice5[0], act=ops.ACT_RELU)
    return [f1, f2, f3, f4, f5]


def lpips_loss(ctx: Ctx, L: LPIPSAlex, real: Tensor, fake: Var, weight: float) -> Tensor:
    """weight * mean_n LPIPS(real_n, fake_n); the gradient w.r.t. `fake` is seeded on the tape (real is a constant)."""
    with torch.no_grad():
        f0s = [f.data for f in features(Ctx([]), L, A.const(real))]
    f1s = features(ctx, L, fake)
    N = real.shape[0]
    total = None
    for k, (f0, f1) in enumerate(zip(f0s, f1s)):
        w = getattr(L, f"lin{k}").model[1].weight.reshape(-1).contiguous()
        f1d = A._dense(f1.data)
        _, Cc, H, W = f1d.shape
        pix = torch.empty((N, H * W), dtype=torch.float32, device=f1d.device)
        df1 = torch.empty_like(f1d)
        check(lib().dcvic_lpips_tap_f32(_p(f0), _p(f1d), _p(w), _p(pix), _p(df1), N, Cc, H * W, C.c_float(weight / N), _stream()), "lpips_tap")
        A.acc(f1, df1)
        v = K.reduce_loss(3, pix, None, weight / (N * H * W))
        total = v if total is None else K.ew(10, None, total, v)
    return total
