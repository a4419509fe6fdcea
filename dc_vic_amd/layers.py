"""Parameter-holding building blocks whose forward passes are HIP kernels.

Each class keeps the parameter names / shapes of the torch.nn layer it stands in for (so the
reference's checkpoints load by key), but `forward` dispatches to the C ABI through dc_vic_amd.ops.
Weights are packed into the MFMA tile layout lazily and re-packed if a parameter changes
(load_state_dict, .to(device)).  Inference only: no autograd graph is built.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops

Tensor = torch.Tensor


class _Packed(nn.Module):
    def __init__(self):
        super().__init__()
        self._plan = None
        self._plan_key = None

    def _key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters(recurse=False))

    def _get_plan(self):
        k = self._key()
        if self._plan is None or self._plan_key != k:
            self._plan = self._build_plan()
            self._plan_key = k
        self._plan.wino = bool(getattr(self, "wino", False))
        self._plan.wino44 = bool(getattr(self, "wino44", False))
        return self._plan

    def _build_plan(self):
        raise NotImplementedError


class Conv2d(_Packed):
    """torch.nn.Conv2d(in, out, k, stride, padding) stand-in (weight [out, in, k, k], bias [out]).

    `asym_pad`: ldm Downsample semantics -- F.pad(x, (0,1,0,1)) then a pad-0 stride-2 conv (model.py:72-76).
    `upsample`: ldm Upsample semantics -- nearest x2 before the conv (model.py:53-57), fused in the loader.
    """

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int = 1, padding: int = 0, bias: bool = True,
                 asym_pad: bool = False, upsample: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_ch, out_ch
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.asym_pad, self.upsample = asym_pad, upsample
        self.wino = False     # allow_winograd(): set only where no integer decision depends on this layer's exact bits
        self.wino44 = False   # allow_winograd(f44=True): F(4x4, 3x3) too -- only AFTER the path's last integer decision
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch, kernel_size, kernel_size), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_ch), requires_grad=False) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_ch * kernel_size * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def _build_plan(self):
        pad = (0, 0) if self.asym_pad else (self.padding, self.padding)
        return ops.ConvPlan(self.weight, self.bias, "conv", stride=self.stride, pad=pad, upsample=self.upsample)

    def forward(self, x, act: int = ops.ACT_NONE, res: Optional[Tensor] = None, affine=None, out: Optional[Tensor] = None,
                gn_stats: bool = False) -> Tensor:
        """`gn_stats`: a GroupNorm over exactly this output follows; its statistics come out of this launch's epilogue when it runs on the
        F(4x4) kernel (take them with `take_gn_part()`, None otherwise -- the GroupNorm then makes its own pass)."""
        plan = self._get_plan()
        out_hw = None
        if self.asym_pad:
            src0 = x if isinstance(x, Tensor) else x[0]
            H, W = src0.shape[2:]
            out_hw = ((H + 1 - self.kernel_size) // self.stride + 1, (W + 1 - self.kernel_size) // self.stride + 1)
        y = plan(x, out=out, act=act, res=res, affine=affine, out_hw=out_hw, gn_stats=gn_stats)
        self._gn_part = plan.last_gn_part
        return y

    def take_gn_part(self):
        p, self._gn_part = getattr(self, "_gn_part", None), None
        return p


def allow_winograd(module: nn.Module, on: bool = True, f44: bool = False) -> nn.Module:
    """Let every Conv2d(k3, s1, p1) under `module` run as Winograd F(2x2, 3x3) (csrc/wino.hip) when the launch is eligible; with
    `f44` also as F(4x4, 3x3) (csrc/wino44.hip: 2.25 instead of 4 multiplies per output, ~3x the rounding error -- 1.5e-6 rms relative
    per layer).  `f44` is passed ONLY by the frozen VQGAN decoder and the SFT fusion blocks, the layers after the estimator's argmax,
    where nothing but the reconstruction's fp tolerance (contract 1e-3, measured in tests/parity_util.py) depends on the bits; never by
    the encoder side, whose bits feed the VQ argmin and the symbol rounding.  `DCVIC_WINO44=0` keeps those layers on F(2x2).

    Winograd re-associates the sum (1e-6 relative against the direct fmaf chain; closer to fp64 than the direct chain).  Callers:
      * the frozen VQGAN decoder and the SFT fusion blocks -- the layers after the path's last integer decision (the estimator
        argmax): bitstreams and indices cannot change, only the reconstruction at the 1e-5 level;
      * the VQGAN ENCODER (vqgan.Encoder, default on, `DCVIC_WINO_ENCODER=0` for the direct kernels) -- upstream of the VQ argmin
        and, through the one-hot feature, of the y symbols.  Deterministic and batch-invariant, so encoder and decoder of THIS
        build always agree; against the oracle / reference it can move a VQ index or a symbol that sits on an fp32 near-tie
        (as any other fp32 summation order would): the parity tests itemise and bound such near-ties, and measured on the test
        set it adds none over the direct kernels.  Its 4-channel conv_out stays direct.
    Hyperprior, CHARM, both ELIC networks and the estimator keep the layer-defined reduction order (never called for them)."""
    for m in module.modules():
        if isinstance(m, Conv2d) and m.kernel_size == 3 and m.stride == 1 and m.padding == 1 and not m.asym_pad:   # (incl. the Upsample convs)
            m.wino = on
            m.wino44 = bool(on and f44)
    return module


class ConvTranspose2d(_Packed):
    """torch.nn.ConvTranspose2d stand-in for the two shapes on the path (weight [in, out, k, k]):
    (k5, s2, p2, output_padding 1) -- elic_autoencoder.py:21-28, minnen20_hyperprior.py:46-48 -- and (k3, s1, p1)."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int, padding: int, output_padding: int = 0):
        super().__init__()
        if (kernel_size, stride, padding, output_padding) not in ((5, 2, 2, 1), (3, 1, 1, 0)):
            raise NotImplementedError("ConvTranspose2d: only (k5,s2,p2,op1) and (k3,s1,p1) are on the DC-VIC path")
        self.in_channels, self.out_channels, self.kernel_size = in_ch, out_ch, kernel_size
        self.weight = nn.Parameter(torch.empty(in_ch, out_ch, kernel_size, kernel_size), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_ch), requires_grad=False)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.zeros_(self.bias)

    def _build_plan(self):
        return ops.ConvPlan(self.weight, self.bias, "convT")

    def forward(self, x, act: int = ops.ACT_NONE, out: Optional[Tensor] = None) -> Tensor:
        return self._get_plan()(x, out=out, act=act)


class Linear(_Packed):
    """torch.nn.Linear stand-in applied to NCHW maps as a 1x1 convolution (weight [out, in])."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features), requires_grad=False) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            nn.init.zeros_(self.bias)

    def _build_plan(self):
        return ops.ConvPlan(self.weight, self.bias, "conv")

    def forward(self, x: Tensor, act: int = ops.ACT_NONE, res: Optional[Tensor] = None) -> Tensor:
        """x: [N, in, H, W] map (a [B, in] matrix is passed as [B, in, 1, 1])."""
        return self._get_plan()(x, act=act, res=res)


class GroupNorm(nn.Module):
    """GroupNorm(32, C, eps 1e-6, affine) (+ fused activation)."""

    def __init__(self, num_channels: int, num_groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.num_groups, self.eps = num_groups, eps
        self.weight = nn.Parameter(torch.ones(num_channels), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(num_channels), requires_grad=False)

    def forward(self, x: Tensor, act: int = ops.ACT_NONE, out: Optional[Tensor] = None, part=None) -> Tensor:
        return ops.groupnorm(x, self.weight, self.bias, self.num_groups, self.eps, act, out, part=part)


class LayerNormC(nn.Module):
    """nn.LayerNorm(C) of the token view, evaluated on the NCHW map."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(dim), requires_grad=False)

    def forward(self, x: Tensor) -> Tensor:
        return ops.layernorm_c(x, self.weight, self.bias, self.eps)


class Act(nn.Module):
    """Placeholder that keeps nn.Sequential indices aligned with the reference (activations are fused)."""

    def forward(self, x):
        raise RuntimeError("activation placeholders are fused into the neighbouring kernel")
