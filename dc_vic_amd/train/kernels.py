"""ctypes wrappers of the training-step kernels (csrc/train.hip, include/dcvic.h "Training step").  torch only
supplies device memory; every computation is a HIP kernel launch on the current stream."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import ops
from .._lib import check, lib
from ..ops import _bs, _chk4, _p, _stream

Tensor = torch.Tensor
_WS = {}


def _workspace(n_floats: int, device, tag: str = "f") -> Tensor:
    """Grow-only scratch buffers (one per tag and device), reused across launches on the same stream."""
    key = (tag, str(device))
    t = _WS.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.empty(max(n_floats, 1 << 16), dtype=torch.float32, device=device)
        _WS[key] = t
    return t


def conv_wgrad(G: Tensor, X: Tensor, dW: Tensor, KH: int, KW: int, stride: int, pad: int, accumulate: bool = True) -> None:
    """dW[m][c][ky][kx] (+)= sum G[n][m][oy][ox] * X[n][c][oy*s+ky-p][ox*s+kx-p]; dW contiguous [M, Cx, KH, KW]."""
    N, M, Hg, Wg = _chk4(G, "wgrad G")
    Nx, Cx, Hx, Wx = _chk4(X, "wgrad X")
    if N != Nx or not dW.is_contiguous() or dW.numel() != M * Cx * KH * KW:
        raise ValueError(f"conv_wgrad: shapes G{tuple(G.shape)} X{tuple(X.shape)} dW{tuple(dW.shape)}")
    need = int(lib().dcvic_conv_wgrad_workspace_floats(N, M, Cx, KH, KW, Hg, None))
    ws = _workspace(need, G.device, "wgrad")
    check(lib().dcvic_conv_wgrad_f32(_p(G), C.c_longlong(_bs(G)), M, Hg, Wg, _p(X), C.c_longlong(_bs(X)), Cx, Hx, Wx, N, KH, KW, stride,
                                     pad, pad, _p(dW), 1 if accumulate else 0, _p(ws), _stream()), "conv_wgrad")


def chan_reduce(a: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """[N, C, H, W] -> [N, C]: sum_p a (* b)."""
    N, Cc, H, W = _chk4(a, "chan_reduce a")
    if b is not None:
        _chk4(b, "chan_reduce b")
    out = torch.empty((N, Cc), dtype=torch.float32, device=a.device)
    check(lib().dcvic_chan_reduce_f32(_p(a), C.c_longlong(_bs(a)), _p(b), C.c_longlong(_bs(b) if b is not None else 0), _p(out), N, Cc, H * W,
                                      _stream()), "chan_reduce")
    return out


def sum_rows(x: Tensor, out: Tensor, accumulate: bool) -> None:
    """out[j] (+)= sum_i x[i][j]  (x: [rows, len] contiguous)."""
    rows = x.shape[0]
    ln = x.numel() // rows
    assert x.is_contiguous() and out.is_contiguous() and out.numel() == ln
    check(lib().dcvic_sum_rows_f32(_p(x), _p(out), rows, C.c_longlong(ln), 1 if accumulate else 0, _stream()), "sum_rows")


def ew(op: int, g: Optional[Tensor], a: Optional[Tensor] = None, b: Optional[Tensor] = None, w: float = 1.0, act: int = 0,
       out: Optional[Tensor] = None, vec_bs: int = 0) -> Tensor:
    ref = g if g is not None else a
    if out is None:
        out = torch.empty(ref.shape, dtype=torch.float32, device=ref.device)
    for t in (g, a if op != 7 else None, b, out):
        if t is not None and not t.is_contiguous():
            raise ValueError("ew_bwd operands must be contiguous")
    Cc = ref.shape[1] if ref.dim() == 4 else 1
    HW = ref.shape[2] * ref.shape[3] if ref.dim() == 4 else 1
    check(lib().dcvic_ew_bwd_f32(op, _p(out), _p(g), _p(a), _p(b), C.c_longlong(ref.numel()), C.c_float(w), act, Cc, HW,
                                 C.c_longlong(vec_bs), _stream()), "ew_bwd")
    return out


def groupnorm_bwd(x: Tensor, dy: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, act: int):
    N, Cc, H, W = _chk4(x, "gn_bwd x")
    _chk4(dy, "gn_bwd dy")
    dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    dg = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
    db = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
    check(lib().dcvic_groupnorm_bwd_f32(_p(x), C.c_longlong(_bs(x)), _p(dy), C.c_longlong(_bs(dy)), _p(dx), C.c_longlong(_bs(dx)), _p(gamma),
                                        _p(beta), _p(dg), _p(db), N, Cc, H * W, groups, C.c_float(eps), act, _stream()), "groupnorm_bwd")
    return dx, dg, db


def layernorm_c_bwd(x: Tensor, dy: Tensor, gamma: Tensor, eps: float):
    N, Cc, H, W = _chk4(x, "ln_bwd x")
    if not (x.is_contiguous() and dy.is_contiguous()):
        raise ValueError("layernorm_c_bwd needs contiguous maps")
    dx = torch.empty_like(x)
    blocks = int(lib().dcvic_layernorm_c_bwd_blocks(N, H * W))
    part = torch.empty((blocks, 2 * Cc), dtype=torch.float32, device=x.device)
    check(lib().dcvic_layernorm_c_bwd_f32(_p(x), _p(dy), _p(dx), _p(gamma), _p(part), N, Cc, H * W, C.c_float(eps), _stream()), "layernorm_c_bwd")
    return dx, part


def softmax_c_bwd(P: Tensor, dP: Tensor, scale: float) -> Tensor:
    N, Cc, Pn = P.shape
    dS = torch.empty_like(P)
    check(lib().dcvic_softmax_c_bwd_f32(_p(P), _p(dP), _p(dS), N, Cc, Pn, C.c_float(scale), _stream()), "softmax_c_bwd")
    return dS


def swin_attn_bwd(qkv: Tensor, dout: Tensor, table: Tensor, dtable: Tensor, heads: int, ws: int, shift: int, accumulate: bool = True) -> Tensor:
    N, C3, H, W = _chk4(qkv, "swin_bwd qkv")
    Cc = C3 // 3
    dqkv = torch.empty_like(qkv)
    nwin = N * (H // ws) * (W // ws)
    wsb = _workspace(nwin * heads * ws ** 4, qkv.device, "swin")
    check(lib().dcvic_swin_attn_bwd_f32(_p(qkv), _p(dout.contiguous()), _p(dqkv), _p(table), _p(dtable), _p(wsb), N, Cc, H, W, heads, ws, shift,
                                        1 if accumulate else 0, _stream()), "swin_attn_bwd")
    return dqkv


def _dws(device) -> Tensor:
    key = ("loss_ws", str(device))
    if key not in _WS:
        _WS[key] = torch.empty(1024, dtype=torch.float64, device=device)
    return _WS[key]


def reduce_loss(kind: int, a: Tensor, b: Optional[Tensor], scale: float, target: int = 0) -> Tensor:
    """0-dim device tensor: scale * sum f(a, b)."""
    if not a.is_contiguous() or (b is not None and not b.is_contiguous()):
        raise ValueError("reduce_loss operands must be contiguous")
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(lib().dcvic_reduce_loss_f32(kind, _p(a), _p(b), C.c_longlong(a.numel()), target, C.c_double(scale), _p(out), _p(_dws(a.device)),
                                      _stream()), "reduce_loss")
    return out


def cross_entropy(logits: Tensor, target: Tensor, w: float, want_grad: bool = True):
    N, Cc, H, W = _chk4(logits, "ce logits")
    if not logits.is_contiguous() or target.dtype != torch.int64 or not target.is_contiguous():
        raise ValueError("cross_entropy: contiguous fp32 logits and int64 targets")
    nll = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib().dcvic_cross_entropy_f32(_p(logits), _p(target), _p(nll), _p(dl), N, Cc, H * W, C.c_float(w), _stream()), "cross_entropy")
    return nll, dl


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, b1: float, b2: float, eps: float, step: int, gscale: Optional[Tensor]) -> None:
    check(lib().dcvic_adam_step_f32(_p(p), _p(g), _p(m), _p(v), C.c_longlong(p.numel()), C.c_float(lr), C.c_float(b1), C.c_float(b2),
                                    C.c_float(eps), step, _p(gscale), _stream()), "adam_step")


def clip_scale(sumsq: Tensor, max_norm: float) -> Tensor:
    out = torch.empty(1, dtype=torch.float32, device=sumsq.device)
    check(lib().dcvic_clip_scale_f32(_p(sumsq), C.c_float(max_norm), _p(out), _stream()), "clip_scale")
    return out


def resample2(x: Tensor, down: bool) -> Tensor:
    N, Cc, H, W = _chk4(x, "resample x")
    if not x.is_contiguous():
        raise ValueError("resample2 needs a contiguous map")
    if down:
        out = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
        check(lib().dcvic_resample2_f32(1, _p(x), _p(out), C.c_longlong(N * Cc), H // 2, W // 2, _stream()), "resample2")
    else:
        out = torch.empty((N, Cc, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        check(lib().dcvic_resample2_f32(0, _p(x), _p(out), C.c_longlong(N * Cc), H, W, _stream()), "resample2")
    return out
