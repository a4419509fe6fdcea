"""Thin Python wrappers over the C ABI (include/dcvic.h).  torch is used only for device memory
and the current HIP stream; every computation below is a hand-written HIP kernel (or the C++ host
entropy coder).  Nothing here falls back to torch ops."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from ._lib import (ACT_GELU, ACT_HALF_TANH, ACT_LRELU02, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH,  # noqa: F401
                   ConvDesc, ConvIO, GemmArgs, check, lib)

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk4(t: Tensor, name: str = "tensor") -> Tuple[int, int, int, int]:
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 4:
        raise ValueError(f"{name}: need a 4-D fp32 device tensor, got {t.dtype} {tuple(t.shape)} {t.device}")
    if t.device.index != torch.cuda.current_device():
        # kernels launch on the CURRENT device's current stream: a tensor of another GPU would be dereferenced by the
        # wrong device.  BaseModel.__init__ / the CLIs make the model's device current; anything else must too.
        raise ValueError(f"{name} lives on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                         f"call torch.cuda.set_device({t.device.index}) (or run under `with torch.cuda.device(...)`) first")
    N, Cc, H, W = t.shape
    st = t.stride()
    if W > 1 and st[3] != 1 or (H > 1 and st[2] != W) or (Cc > 1 and st[1] != H * W):
        raise ValueError(f"{name}: channel planes must be dense NCHW (shape {tuple(t.shape)}, strides {st})")
    return N, Cc, H, W


def _bs(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1] * t.shape[2] * t.shape[3])


def _p(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def new(N, Cc, H, W, like: Tensor) -> Tensor:
    return torch.empty((N, Cc, H, W), dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------------- kernel events
# Per-launch HIP events on the stream the conv kernels run on (torch's current stream), used by
# bench.py to report the dominant kernel's achieved TFLOP/s inside the timed region.
_EVENTS = None
_VARIANT_TILES = {(0, 256): "2, 4, 2, 2", (0, 128): "2, 2, 2, 2", (0, 64): "2, 1, 2, 2", (1, 256): "2, 2, 1, 4", (1, 128): "1, 2, 2, 2",
                  (1, 64): "1, 1, 2, 2", (2, 256): "1, 2, 1, 4", (2, 128): "1, 1, 1, 4", (3, 256): "3, 2, 1, 4", (3, 128): "3, 1, 1, 4"}


def conv_kernel_name(variant: int) -> str:
    """Kernel name as rocprofv3 prints it, from dcvic_conv_last_variant()."""
    if variant == 9000:
        return "void conv3x3_dma_kernel<3, 3, 4, 128>(ConvKArgs)"
    if variant == 9001:
        return "void conv3x3_dma_kernel<2, 2, 8, 128>(ConvKArgs)"
    if variant == 9003:
        return "void conv3x3_dma_kernel<3, 3, 4, 96>(ConvKArgs)"
    if variant == 9100:
        return "void conv3x3_wino_kernel<0>(ConvKArgs)"
    if variant == 9101:
        return "conv3x3_wino_ups_kernel(ConvKArgs)"
    if variant == 9104:
        return "conv3x3_wino44_kernel(ConvKArgs)"
    if 9200 <= variant < 9210:
        return f"void thin_cout_kernel<{variant - 9200}>(ThinArgs)"
    if 9210 <= variant < 9220:
        return f"void thin_cin_kernel<{variant - 9210}>(ThinArgs)"
    if 7000 <= variant < 8000:                       # conv1x1.hip: 7000 + cls
        return f"void conv1x1_dma_kernel<{(128, 64, 32, 96)[variant - 7000]}>(ConvKArgs)"
    if 8500 <= variant < 9000:                       # conv_async16.hip: 8500 + cls*100 + P/32 (tiles in units of 16)
        cls, p32 = divmod(variant - 8500, 100)
        t16 = {(0, 64): "4, 2, 2, 2", (1, 128): "2, 4, 2, 2", (1, 64): "2, 2, 2, 2", (2, 256): "2, 4, 1, 4", (2, 128): "2, 2, 1, 4",
               (2, 64): "2, 1, 1, 4", (2, 32): "1, 1, 2, 2"}.get((cls, p32 * 32), "?")
        return f"void conv_mfma_async16_kernel<{t16}>(ConvKArgs, int, int)"
    if 8000 <= variant < 9000:                       # conv_async.hip: 8000 + cls*100 + P/32
        cls, p32 = divmod(variant - 8000, 100)
        return f"void conv_mfma_async_kernel<{_VARIANT_TILES.get((cls, p32 * 32), '?')}>(ConvKArgs, int, int)"
    cls, rest = divmod(variant, 1000)
    P, ups = rest - (rest & 1), rest & 1
    return f"void conv_mfma_kernel<{_VARIANT_TILES.get((cls, P), '?')}, {'true' if ups else 'false'}>(ConvKArgs)"


# MFMA flops a kernel EXECUTES per algorithmic (direct-convolution 2*MAC) flop: Winograd F(2x2,3x3) issues 16 multiplies per 2x2
# outputs and input channel instead of 36, the upsample-structured form 9 of 36, F(4x4,3x3) 36 per 4x4 outputs instead of 144
EXECUTED_FRACTION = {9100: 16.0 / 36.0, 9101: 9.0 / 36.0, 9104: 36.0 / 144.0}


def kernel_events_start() -> None:
    global _EVENTS
    _EVENTS = []


def kernel_events_stop():
    """-> {kernel name: {kernel, launches, flops (algorithmic 2*MAC), exec_flops (MFMA flops issued), bytes (algorithmic HBM bytes:
    one read of the input, one write of the output, the weights; residual reads not counted), time_s}}; call after a device sync."""
    global _EVENTS
    ev, _EVENTS = _EVENTS, None
    out = {}
    shapes = {}
    for cfg, flops, e0, e1, shp in ev or []:
        name = conv_kernel_name(cfg)
        d = out.setdefault(name, {"kernel": name, "launches": 0, "flops": 0.0, "exec_flops": 0.0, "bytes": 0.0, "time_s": 0.0})
        t = e0.elapsed_time(e1) * 1e-3
        d["launches"] += 1
        d["flops"] += flops
        Cin, Cout, T, st, ups, H, W, N = shp
        Ho, Wo = (2 * H, 2 * W) if ups else (-(-H // st), -(-W // st))
        d["bytes"] += 4.0 * (N * Cin * H * W + N * Cout * Ho * Wo + Cout * Cin * T)
        d["exec_flops"] += flops * EXECUTED_FRACTION.get(cfg, 1.0)
        d["time_s"] += t
        sd = shapes.setdefault((cfg,) + shp, [0, 0.0, 0.0])
        sd[0] += 1; sd[1] += flops; sd[2] += t
    global LAST_SHAPE_STATS
    LAST_SHAPE_STATS = shapes
    return out


LAST_SHAPE_STATS = {}


def shape_stats_report() -> str:
    """Per conv shape: launches, total ms, achieved TFLOP/s (diagnostic for kernel tuning)."""
    rows = sorted(LAST_SHAPE_STATS.items(), key=lambda kv: -kv[1][2])
    lines = ["cfg Cin Cout T s ups H W N | launches ms TFLOP/s"]
    for k, (n, fl, t) in rows:
        lines.append(f"{k} | {n} {t * 1e3:.2f} {fl / t / 1e12:.1f}")
    return "\n".join(lines)


# ------------------------------------------------------------------------------------------- conv
# Winograd F(2x2,3x3) for the layers that opted in (ConvPlan.wino): DCVIC_WINO=0 keeps every layer on the direct kernels
WINO_ENABLED = os.environ.get("DCVIC_WINO", "1") != "0"
GN_FUSED_STATS = os.environ.get("DCVIC_GN_FUSED", "1") != "0"   # GroupNorm statistics from the producing F(4x4) convolution's epilogue
THIN_MIN_PIXELS = int(os.environ.get("DCVIC_THIN_MIN_PIXELS", "16384"))   # full-resolution maps only (the 4-channel latent convs at 1/8 stay on the MFMA kernels)
THIN_ENABLED = os.environ.get("DCVIC_THIN", "1") != "0"       # VALU kernels for the 3 -> 128 / 128 -> 3 layers (csrc/thin.hip)
WINO44_MIN_BLOCKS = int(os.environ.get("DCVIC_WINO44_MIN_BLOCKS", "16"))   # workgroup tiles PER IMAGE below which F(2x2) / direct run
WINO44_ENABLED = os.environ.get("DCVIC_WINO44", "1") != "0"   # F(4x4,3x3) for the layers that opted in (ConvPlan.wino44)
WINO_MIN_BLOCKS = int(os.environ.get("DCVIC_WINO_MIN_BLOCKS", "16"))   # workgroups PER IMAGE below which the direct kernels run


class ConvPlan:
    """A convolution layer bound to the C ABI: descriptor(s) + packed weights.

    kind 'conv'  : Conv2d(k, stride, padding given as (pad_t, pad_l)), optional nearest-x2 input
    kind 'convT' : ConvTranspose2d(k=5,s=2,p=2,op=1) (four output phases) or (k=3,s=1,p=1)
    """

    wino = False          # set by the owner: Winograd F(2x2,3x3) allowed (no integer decision downstream of this layer)
    wino44 = False        # set by the owner: F(4x4,3x3) allowed too -- post-argmax layers only (3x the F(2x2) rounding error)
    _wino_pack = None
    _wino44_pack = None
    last_gn_part = None   # (partial statistics [N, Cout, n_pt, 2], n_pt) written by the last call when gn_stats was asked AND the F(4x4) kernel ran
    _wino_ups_pack = None

    def __init__(self, weight: Tensor, bias: Optional[Tensor], kind: str = "conv", stride: int = 1,
                 pad: Tuple[int, int] = (0, 0), upsample: bool = False):
        self.kind = kind
        self.stride = stride
        self.pad = pad
        self.upsample = upsample
        self.bias = bias.detach().contiguous() if bias is not None else None
        w = weight.detach().contiguous()
        if w.dim() == 2:
            w = w.view(w.shape[0], w.shape[1], 1, 1)
        self._w = w
        self.phases: List[list] = []      # [desc, {tile class: packed weights}, py, px]
        self.ups_phases = False
        if kind == "conv" and upsample and tuple(w.shape[2:]) == (3, 3) and pad == (1, 1) and stride == 1 \
                and os.environ.get("DCVIC_UPS_PHASES", "1") != "0":
            # nearest-x2 + conv3x3 == four 2x2 sub-pixel convolutions on the low-res input (4/9 of the flops):
            # output row 2m+py reads low-res rows {m-1, m} (py=0) or {m, m+1} (py=1) with the taps that land on
            # the same low-res row pre-added.  Zero padding is preserved (row -1 / row H are out of the image).
            self.Cout, self.Cin, self.KH, self.KW = w.shape
            self.ups_phases = True
            self._wphase = []
            for py in (0, 1):
                rows = [w[:, :, 0], w[:, :, 1] + w[:, :, 2]] if py == 0 else [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]
                for px in (0, 1):
                    cols = []
                    for r in rows:       # r: [Cout, Cin, 3]
                        cols.append(torch.stack([r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if px == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]], dim=-1))
                    weff = torch.stack(cols, dim=2).contiguous()    # [Cout, Cin, 2, 2]
                    d = ConvDesc()
                    check(lib().dcvic_conv_desc_init(C.byref(d), self.Cin, self.Cout, 2, 2, 1, 1 - py, 1 - px, 0), "conv_desc_init")
                    self.phases.append([d, {}, py, px])
                    self._wphase.append(weff)
        elif kind == "conv":
            self.Cout, self.Cin, self.KH, self.KW = w.shape
            d = ConvDesc()
            check(lib().dcvic_conv_desc_init(C.byref(d), self.Cin, self.Cout, self.KH, self.KW, stride, pad[0], pad[1],
                                             1 if upsample else 0), "conv_desc_init")
            self.phases.append([d, {}, 0, 0])
        elif kind == "convT":
            self.Cin, self.Cout, self.KH, self.KW = w.shape
            k = self.KH
            if k == 3:
                d = ConvDesc()
                check(lib().dcvic_convT_phase_desc(C.byref(d), self.Cin, self.Cout, 3, 0, 0), "convT_phase_desc")
                self.phases.append([d, {}, 0, 0])
            else:
                for py in (0, 1):
                    for px in (0, 1):
                        d = ConvDesc()
                        check(lib().dcvic_convT_phase_desc(C.byref(d), self.Cin, self.Cout, k, py, px), "convT_phase_desc")
                        self.phases.append([d, {}, py, px])
        else:
            raise ValueError(kind)

    @classmethod
    def from_phases2(cls, wphases: Sequence[Tensor]) -> "ConvPlan":
        """Four 2x2 sub-pixel convolutions writing the (py, px) phases of a x2-sized output (order (0,0), (0,1), (1,0), (1,1));
        phase (py, px) reads low-resolution rows {m-1, m} (py = 0) or {m, m+1} (py = 1), same for columns.  This is the
        shape of `nearest x2 + conv3x3` AND of ConvTranspose2d(k4, s2, p1) -- the data gradient of a Conv2d(k4, s2, p1)."""
        self = cls.__new__(cls)
        w0 = wphases[0]
        self.kind, self.stride, self.pad, self.upsample = "conv", 1, (1, 1), True
        self.bias = None
        self._w = None
        self.Cout, self.Cin, self.KH, self.KW = w0.shape[0], w0.shape[1], 3, 3
        self.ups_phases = True
        self.phases, self._wphase = [], []
        for (py, px), wp in zip(((0, 0), (0, 1), (1, 0), (1, 1)), wphases):
            d = ConvDesc()
            check(lib().dcvic_conv_desc_init(C.byref(d), self.Cin, self.Cout, 2, 2, 1, 1 - py, 1 - px, 0), "conv_desc_init")
            self.phases.append([d, {}, py, px])
            self._wphase.append(wp.detach().contiguous())
        return self

    def _wino_ok(self, srcs, N: int, H: int, W: int) -> bool:
        """Winograd eligibility: Conv2d(k3, s1, p1), 8-channel-aligned sources, even width, and a map that fills its
        64-channel x 8 x 32-pixel workgroup tiles.  A function of the LAYER and the IMAGE size only, never of N: a
        reconstruction must not depend on the batch it was decoded in."""
        if self.kind != "conv" or self.upsample or self.ups_phases or self.stride != 1 or self.pad != (1, 1) \
                or (self.KH, self.KW) != (3, 3) or self._w is None:
            return False
        if (W & 3) or self.Cout < 48 or any(s.shape[1] % 8 or s.data_ptr() % 16 or (s.shape[0] > 1 and s.stride(0) % 4) for s in srcs):
            return False
        ty, tx = (H + 7) // 8, (W + 31) // 32
        if H * W < 0.6 * (ty * 8 * tx * 32):
            return False
        return ty * tx * ((self.Cout + 63) // 64) >= WINO_MIN_BLOCKS

    def _wino44_ok(self, srcs, N: int, H: int, W: int) -> bool:
        """F(4x4, 3x3) eligibility: Conv2d(k3, s1, p1), every source a multiple of 8 channels, width % 4 == 0, and a map that fills its
        64-channel x 16 x 32-pixel workgroup tiles.  A function of the layer and the image size only, never of N."""
        if self.kind != "conv" or self.upsample or self.ups_phases or self.stride != 1 or self.pad != (1, 1) \
                or (self.KH, self.KW) != (3, 3) or self._w is None:
            return False
        if (W & 3) or self.Cout < 48 or any(s.shape[1] % 8 or s.data_ptr() % 16 or (s.shape[0] > 1 and s.stride(0) % 4) for s in srcs):
            return False
        ty, tx = (H + 15) // 16, (W + 31) // 32
        if H * W < 0.6 * (ty * 16 * tx * 32):
            return False
        return ty * tx * ((self.Cout + 63) // 64) >= WINO44_MIN_BLOCKS

    def _wino_ups_ok(self, srcs, H: int, W: int) -> bool:
        """Eligibility of the upsample-fused Winograd (input H x W, output 2H x 2W): as _wino_ok, on the output's tile grid."""
        if (W & 3) or self.Cout < 48 or any(s.shape[1] % 8 or s.data_ptr() % 16 or (s.shape[0] > 1 and s.stride(0) % 4) for s in srcs):
            return False
        ty, tx = (2 * H + 7) // 8, (2 * W + 31) // 32
        if 4 * H * W < 0.6 * (ty * 8 * tx * 32):
            return False
        return ty * tx * ((self.Cout + 63) // 64) >= WINO_MIN_BLOCKS

    @staticmethod
    def _pack(d: ConvDesc, w: Tensor) -> Tensor:
        nbytes = lib().dcvic_conv_packed_bytes(C.byref(d))
        packed = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        check(lib().dcvic_conv_pack_f32(C.byref(d), _p(w), _p(packed), _stream()), "conv_pack")
        return packed

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        if self.kind == "convT":
            return (H, W) if self.KH == 3 else (2 * H, 2 * W)
        if self.upsample:
            H, W = 2 * H, 2 * W
        return H, W  # overridden by the caller for strided / valid convs

    def __call__(self, srcs, out: Optional[Tensor] = None, act: int = ACT_NONE, res: Optional[Tensor] = None,
                 affine: Optional[Tuple[Tensor, Tensor]] = None, out_hw: Optional[Tuple[int, int]] = None,
                 init: Optional[Tensor] = None, use_bias: bool = True, gn_stats: bool = False) -> Tensor:
        """`gn_stats`: the caller's next op is a GroupNorm over exactly this output; when the launch runs on the F(4x4) kernel its epilogue
        also writes the GroupNorm partial sums (self.last_gn_part, else None) and the GroupNorm skips its statistics pass."""
        self.last_gn_part = None
        if isinstance(srcs, Tensor):
            srcs = [srcs]
        N, _, H, W = _chk4(srcs[0], "conv src0")
        if self.kind == "conv":
            if out_hw is None:
                Hi, Wi = (2 * H, 2 * W) if self.upsample else (H, W)
                # torch semantics with symmetric padding (pad_t == pad_b)
                Ho = (Hi + 2 * self.pad[0] - self.KH) // self.stride + 1
                Wo = (Wi + 2 * self.pad[1] - self.KW) // self.stride + 1
            else:
                Ho, Wo = out_hw
            Hf, Wf = Ho, Wo
        else:
            Hf, Wf = (H, W) if self.KH == 3 else (2 * H, 2 * W)
        if out is None:
            out = new(N, self.Cout, Hf, Wf, srcs[0])
        No, Co, Hc, Wc = _chk4(out, "conv out")
        if (No, Co, Hc, Wc) != (N, self.Cout, Hf, Wf):
            raise ValueError(f"conv out shape {tuple(out.shape)} != {(N, self.Cout, Hf, Wf)}")
        io = ConvIO()
        io.N, io.H, io.W = N, H, W
        io.Hfull, io.Wfull = Hf, Wf
        io.n_src = len(srcs)
        for i, s in enumerate(srcs):
            n_, c_, h_, w_ = _chk4(s, f"conv src{i}")
            if (n_, h_, w_) != (N, H, W):
                raise ValueError("conv sources disagree in shape")
            io.src[i].ptr = s.data_ptr(); io.src[i].C = c_; io.src[i].batch_stride = _bs(s)
        io.out = out.data_ptr(); io.out_batch_stride = _bs(out)
        io.bias = self.bias.data_ptr() if (self.bias is not None and use_bias) else None
        io.act = act
        if init is not None:
            if tuple(init.shape) != tuple(out.shape):
                raise ValueError("conv init shape mismatch")
            _chk4(init, "conv init")
            io.init = init.data_ptr(); io.init_batch_stride = _bs(init)
        if res is not None:
            if tuple(res.shape) != tuple(out.shape):
                raise ValueError("conv residual shape mismatch")
            _chk4(res, "conv res")
            io.res = res.data_ptr(); io.res_batch_stride = _bs(res)
        if affine is not None:
            sc, sh = affine
            if sc.shape[-1] != self.Cout or sc.shape != sh.shape or not sc.is_contiguous() or not sh.is_contiguous():
                raise ValueError("conv affine vectors must be contiguous [B, Cout]")
            io.aff_scale = sc.data_ptr(); io.aff_shift = sh.data_ptr()
            io.aff_batch_stride = self.Cout if sc.shape[0] > 1 else 0
            if sc.shape[0] not in (1, N):
                raise ValueError("conv affine batch must be 1 or N")
        st = _stream()
        if THIN_ENABLED and self.kind == "conv" and not self.upsample and not self.ups_phases and self.stride == 1 and self.pad == (1, 1) \
                and (self.KH, self.KW) == (3, 3) and self._w is not None and len(srcs) == 1 and init is None and affine is None \
                and (out_hw is None or tuple(out_hw) == (H, W)) and H * W >= THIN_MIN_PIXELS \
                and lib().dcvic_conv3x3_thin_applies(self.Cin, self.Cout):
            # a thin layer (VQGAN conv_in 3 -> 128, conv_out 128 -> 3): HBM-bound VALU kernel, bit-identical to the MFMA kernels' order
            io.Hout, io.Wout = Hf, Wf
            io.osy = io.osx = 1
            io.ooy = io.oox = 0
            var = 9200 + self.Cout if self.Cout <= 4 and self.Cin >= 8 else 9210 + self.Cin
            if _EVENTS is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib().dcvic_conv3x3_thin_f32(_p(self._w), self.Cin, self.Cout, C.byref(io), st), "conv3x3_thin")
                e1.record()
                _EVENTS.append((var, 2.0 * N * H * W * self.Cout * self.Cin * 9, e0, e1, (self.Cin, self.Cout, 9, 1, 0, H, W, N)))
            else:
                check(lib().dcvic_conv3x3_thin_f32(_p(self._w), self.Cin, self.Cout, C.byref(io), st), "conv3x3_thin")
            return out
        if self.wino44 and WINO44_ENABLED and self.wino and WINO_ENABLED and not self.ups_phases and not self.upsample and init is None and affine is None \
                and act in (ACT_NONE, ACT_RELU, ACT_LRELU02) \
                and (self.wino44 == "force" or self._wino44_ok(srcs, N, H, W)) \
                and out.data_ptr() % 16 == 0 and _bs(out) % 4 == 0 and (res is None or (res.data_ptr() % 16 == 0 and _bs(res) % 4 == 0)):
            if self._wino44_pack is None:
                nbytes = lib().dcvic_wino44_packed_bytes(self.Cin, self.Cout)
                self._wino44_pack = torch.empty(nbytes // 4, dtype=torch.float32, device=self._w.device)
                check(lib().dcvic_wino44_pack_f32(_p(self._w), _p(self._wino44_pack), self.Cin, self.Cout, st), "wino44_pack")
            io.Hout, io.Wout = Hf, Wf
            io.osy = io.osx = 1
            io.ooy = io.oox = 0
            part = None
            if gn_stats and GN_FUSED_STATS:
                n_pt = int(lib().dcvic_wino44_stats_tiles(H, W))
                part = torch.empty((N, self.Cout, n_pt, 2), dtype=torch.float32, device=out.device)
                self.last_gn_part = (part, n_pt)

            def launch():
                if part is None:
                    check(lib().dcvic_conv3x3_wino44_f32(self.Cin, self.Cout, _p(self._wino44_pack), C.byref(io), st), "conv3x3_wino44")
                else:
                    check(lib().dcvic_conv3x3_wino44_stats_f32(self.Cin, self.Cout, _p(self._wino44_pack), C.byref(io), _p(part), st), "conv3x3_wino44_stats")
            if _EVENTS is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                launch()
                e1.record()
                _EVENTS.append((9104, 2.0 * N * H * W * self.Cout * self.Cin * 9, e0, e1, (self.Cin, self.Cout, 9, 1, 0, H, W, N)))
            else:
                launch()
            return out
        if self.wino and WINO_ENABLED and not self.ups_phases and not self.upsample and init is None and affine is None \
                and (self.wino == "force" or self._wino_ok(srcs, N, H, W)) \
                and out.data_ptr() % 16 == 0 and _bs(out) % 4 == 0 and (res is None or (res.data_ptr() % 16 == 0 and _bs(res) % 4 == 0)):
            if self._wino_pack is None:
                nbytes = lib().dcvic_wino_packed_bytes(self.Cin, self.Cout)
                self._wino_pack = torch.empty(nbytes // 4, dtype=torch.float32, device=self._w.device)
                check(lib().dcvic_wino_pack_f32(_p(self._w), _p(self._wino_pack), self.Cin, self.Cout, st), "wino_pack")
            io.Hout, io.Wout = Hf, Wf
            io.osy = io.osx = 1
            io.ooy = io.oox = 0
            if _EVENTS is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib().dcvic_conv3x3_wino_f32(self.Cin, self.Cout, _p(self._wino_pack), C.byref(io), st), "conv3x3_wino")
                e1.record()
                _EVENTS.append((9100, 2.0 * N * H * W * self.Cout * self.Cin * 9, e0, e1, (self.Cin, self.Cout, 9, 1, 0, H, W, N)))
            else:
                check(lib().dcvic_conv3x3_wino_f32(self.Cin, self.Cout, _p(self._wino_pack), C.byref(io), st), "conv3x3_wino")
            return out
        if self.wino and WINO_ENABLED and self.ups_phases and self._w is not None and init is None and affine is None \
                and (self.wino == "force" or self._wino_ups_ok(srcs, H, W)) \
                and out.data_ptr() % 16 == 0 and _bs(out) % 4 == 0 and (res is None or (res.data_ptr() % 16 == 0 and _bs(res) % 4 == 0)):
            # nearest x2 + conv3x3 as the 9-position structured Winograd (csrc/wino.hip) instead of four 2x2 phase convolutions
            if self._wino_ups_pack is None:
                nbytes = lib().dcvic_wino_ups_packed_bytes(self.Cin, self.Cout)
                self._wino_ups_pack = torch.empty(nbytes // 4, dtype=torch.float32, device=self._w.device)
                check(lib().dcvic_wino_ups_pack_f32(_p(self._w), _p(self._wino_ups_pack), self.Cin, self.Cout, st), "wino_ups_pack")
            io.Hout, io.Wout = Hf, Wf
            io.osy = io.osx = 1
            io.ooy = io.oox = 0
            if _EVENTS is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib().dcvic_conv3x3_wino_ups_f32(self.Cin, self.Cout, _p(self._wino_ups_pack), C.byref(io), st), "conv3x3_wino_ups")
                e1.record()
                _EVENTS.append((9101, 2.0 * N * Hf * Wf * self.Cout * self.Cin * 9, e0, e1, (self.Cin, self.Cout, 9, 1, 1, H, W, N)))
            else:
                check(lib().dcvic_conv3x3_wino_ups_f32(self.Cin, self.Cout, _p(self._wino_ups_pack), C.byref(io), st), "conv3x3_wino_ups")
            return out
        for phi, ph in enumerate(self.phases):
            d, packs, py, px = ph
            if self.ups_phases or (self.kind == "convT" and self.KH == 5):
                io.Hout, io.Wout = H, W
                io.osy = io.osx = 2
                io.ooy, io.oox = py, px
            else:
                io.Hout, io.Wout = Hf, Wf
                io.osy = io.osx = 1
                io.ooy = io.oox = 0
            cls = lib().dcvic_conv_select_class(C.byref(d), N, io.Hout, io.Wout)
            if cls < 0:
                check(cls, "conv_select_class")
            d.cfg = cls
            packed = packs.get(cls)
            if packed is None:
                packed = packs[cls] = self._pack(d, self._wphase[phi] if self.ups_phases else self._w)
            if _EVENTS is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib().dcvic_conv2d_f32(C.byref(d), _p(packed), C.byref(io), st), "conv2d")
                e1.record()
                _EVENTS.append((int(lib().dcvic_conv_last_variant()), 2.0 * N * io.Hout * io.Wout * self.Cout * self.Cin * int(d.T), e0, e1,
                                (self.Cin, self.Cout, int(d.T), self.stride, int(self.upsample), H, W, N)))
            else:
                check(lib().dcvic_conv2d_f32(C.byref(d), _p(packed), C.byref(io), st), "conv2d")
        return out


# ------------------------------------------------------------------------------------------- gemm
def bgemm(A: Tensor, a_strides, B: Tensor, b_strides, out: Tensor, c_strides, batch, M, N, K, alpha=1.0):
    g = GemmArgs()
    g.batch, g.M, g.N, g.K = batch, M, N, K
    g.A = A.data_ptr(); g.a_bs, g.a_ms, g.a_ks = a_strides
    g.B = B.data_ptr(); g.b_bs, g.b_ks, g.b_ns = b_strides
    g.C = out.data_ptr(); g.c_bs, g.c_ms = c_strides
    g.alpha = alpha
    check(lib().dcvic_bgemm_f32(C.byref(g), _stream()), "bgemm")
    return out


def attn_fused(qkv: Tensor, Cc: int, out: Optional[Tensor] = None, force_nw: int = 0) -> Tensor:
    """Single-head attention over all H*W positions (ldm AttnBlock, model.py:186-196) from a [N, 3C, H, W] map holding
    q | k | v in its channel thirds -> [N, C, H, W].  Scores stay on chip (online softmax)."""
    N, C3, H, W = _chk4(qkv, "attn qkv")
    if C3 != 3 * Cc or not qkv.is_contiguous():
        raise ValueError("attn_fused needs a contiguous [N, 3C, H, W] q|k|v map")
    HW = H * W
    if out is None:
        out = new(N, Cc, H, W, qkv)
    _chk4(out, "attn out")
    base = qkv.data_ptr()
    check(lib().dcvic_attn_fused_f32(C.c_void_p(base), C.c_void_p(base + 4 * Cc * HW), C.c_void_p(base + 8 * Cc * HW),
                                     C.c_longlong(3 * Cc * HW), _p(out), C.c_longlong(_bs(out)), N, Cc, HW,
                                     C.c_float(float(int(Cc) ** (-0.5))), force_nw, _stream()), "attn_fused")
    return out


# ------------------------------------------------------------------------------------------- norms
def groupnorm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 32, eps: float = 1e-6, act: int = ACT_NONE,
              out: Optional[Tensor] = None, part=None) -> Tensor:
    """`part` = (partial statistics [N, C, n_pt, 2], n_pt) of exactly `x`, from the convolution that produced it (ConvPlan.last_gn_part)."""
    N, Cc, H, W = _chk4(x, "groupnorm x")
    if out is None:
        out = new(N, Cc, H, W, x)
    _chk4(out, "groupnorm out")
    if part is not None:
        pt, n_pt = part
        if tuple(pt.shape) != (N, Cc, n_pt, 2) or not pt.is_contiguous():
            raise ValueError(f"groupnorm: partial statistics {tuple(pt.shape)} do not belong to a {tuple(x.shape)} map")
        check(lib().dcvic_groupnorm_part_f32(_p(x), C.c_longlong(_bs(x)), _p(out), C.c_longlong(_bs(out)), _p(gamma), _p(beta),
                                             N, Cc, H * W, groups, C.c_float(eps), act, _p(pt), n_pt, _stream()), "groupnorm_part")
        return out
    check(lib().dcvic_groupnorm_f32(_p(x), C.c_longlong(_bs(x)), _p(out), C.c_longlong(_bs(out)), _p(gamma), _p(beta),
                                    N, Cc, H * W, groups, C.c_float(eps), act, _stream()), "groupnorm")
    return out


def layernorm_c(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    N, Cc, H, W = _chk4(x, "layernorm x")
    if not x.is_contiguous():
        raise ValueError("layernorm_c needs a contiguous map")
    out = torch.empty_like(x)
    check(lib().dcvic_layernorm_c_f32(_p(x), _p(out), _p(gamma), _p(beta), N, Cc, H * W, C.c_float(eps), _stream()), "layernorm_c")
    return out


def softmax_c_(x: Tensor, N: int, Cc: int, P: int) -> Tensor:
    check(lib().dcvic_softmax_c_f32(_p(x), N, Cc, P, _stream()), "softmax_c")
    return x


def swin_attn(qkv: Tensor, bias_table: Tensor, heads: int, ws: int, shift: int) -> Tensor:
    N, C3, H, W = _chk4(qkv, "swin qkv")
    if not qkv.is_contiguous():
        raise ValueError("swin_attn needs a contiguous qkv map")
    Cc = C3 // 3
    out = new(N, Cc, H, W, qkv)
    check(lib().dcvic_swin_attn_f32(_p(qkv), _p(out), _p(bias_table), N, Cc, H, W, heads, ws, shift, _stream()), "swin_attn")
    return out


# ------------------------------------------------------------------------------------------- elementwise
def _ew(op: int, a: Tensor, b: Optional[Tensor], c: Optional[Tensor], w: float = 1.0, act: int = ACT_NONE,
        out: Optional[Tensor] = None) -> Tensor:
    N, Cc, H, W = _chk4(a, "ew a")
    for t, nm in ((b, "ew b"), (c, "ew c")):
        if t is not None:
            if tuple(t.shape) != tuple(a.shape):
                raise ValueError(f"{nm} shape mismatch")
            _chk4(t, nm)
    if out is None:
        out = new(N, Cc, H, W, a)
    check(lib().dcvic_ew_f32(op, _p(out), C.c_longlong(_bs(out)), _p(a), C.c_longlong(_bs(a)),
                             _p(b), C.c_longlong(_bs(b) if b is not None else 0),
                             _p(c), C.c_longlong(_bs(c) if c is not None else 0),
                             N, Cc, H * W, C.c_float(w), act, _stream()), "ew")
    return out


def add(a: Tensor, b: Tensor, out=None) -> Tensor:
    return _ew(0, a, b, None, out=out)


def add_mul_sigmoid(a: Tensor, b: Tensor, c: Tensor, out=None) -> Tensor:
    """a + b * sigmoid(c)   (ChengNLAM, cheng_nlam.py:23-27)"""
    return _ew(1, a, b, c, out=out)


def sft(a: Tensor, scale: Tensor, shift: Tensor, w: float = 1.0, out=None) -> Tensor:
    """a + w * (a * scale + shift)   (FuseSftBlock, codeformer_layers.py:65-66)"""
    return _ew(2, a, scale, shift, w=w, out=out)


def activation(a: Tensor, act: int, out=None) -> Tensor:
    return _ew(4, a, None, None, act=act, out=out)


def chan_affine(x: Tensor, scale: Tensor, shift: Tensor, add_: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """x * (1 + scale[n][c]) + shift[n][c] (+ add)   (BetaScaleShiftModule.forward)"""
    N, Cc, H, W = _chk4(x, "chan_affine x")
    if out is None:
        out = new(N, Cc, H, W, x)
    if scale.shape[-1] != Cc or scale.shape[0] not in (1, N):
        raise ValueError("chan_affine vectors must be [1|N, C]")
    check(lib().dcvic_chan_affine_f32(_p(out), C.c_longlong(_bs(out)), _p(x), C.c_longlong(_bs(x)), _p(scale), _p(shift),
                                      C.c_longlong(Cc if scale.shape[0] > 1 else 0), _p(add_),
                                      C.c_longlong(_bs(add_) if add_ is not None else 0), N, Cc, H * W, _stream()), "chan_affine")
    return out


def copy_planes(dst: Tensor, src: Tensor, copyH: int, copyW: int, reflect: bool = False) -> Tensor:
    N, Cc, dH, dW = _chk4(dst, "copy dst")
    Ns, Cs, sH, sW = _chk4(src, "copy src")
    if (N, Cc) != (Ns, Cs):
        raise ValueError("copy_planes N/C mismatch")
    check(lib().dcvic_copy_planes_f32(_p(dst), C.c_longlong(_bs(dst)), dH, dW, _p(src), C.c_longlong(_bs(src)), sH, sW, N, Cc,
                                      copyH, copyW, 1 if reflect else 0, _stream()), "copy_planes")
    return dst


def pad_reflect(x: Tensor, padH: int, padW: int) -> Tensor:
    """F.pad(x, (0, padW, 0, padH), mode='reflect')   (base_model.py:156-163)"""
    N, Cc, H, W = _chk4(x, "pad x")
    out = new(N, Cc, H + padH, W + padW, x)
    return copy_planes(out, x, H + padH, W + padW, reflect=True)


def crop(x: Tensor, H: int, W: int) -> Tensor:
    N, Cc, _, _ = _chk4(x, "crop x")
    out = new(N, Cc, H, W, x)
    return copy_planes(out, x, H, W, reflect=False)


def copy_window(dst: Tensor, src: Tensor) -> Tensor:
    """dst[...] = src[...] for two equally shaped 4-D views with unit innermost stride (tile cut / stitch)."""
    if tuple(dst.shape) != tuple(src.shape) or dst.stride(3) != 1 or src.stride(3) != 1:
        raise ValueError("copy_window: views must have equal shapes and unit innermost stride")
    N, Cc, h, w = src.shape
    check(lib().dcvic_copy_window_f32(_p(dst), C.c_longlong(dst.stride(0)), C.c_longlong(dst.stride(1)), C.c_longlong(dst.stride(2)),
                                      _p(src), C.c_longlong(src.stride(0)), C.c_longlong(src.stride(1)), C.c_longlong(src.stride(2)),
                                      N, Cc, h, w, _stream()), "copy_window")
    return dst


def absmax(x: Tensor) -> Tensor:
    N, Cc, H, W = _chk4(x, "absmax x")
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    check(lib().dcvic_absmax_f32(_p(x), C.c_longlong(_bs(x)), _p(out), N, C.c_longlong(Cc * H * W), _stream()), "absmax")
    return out


def crop_clamp(x: Tensor, H: int, W: int, want_u8: bool = False):
    N, Cc, Hs, Ws = _chk4(x, "crop_clamp x")
    y = new(N, Cc, H, W, x)
    y8 = torch.empty((N, H, W, Cc), dtype=torch.uint8, device=x.device) if want_u8 else None
    check(lib().dcvic_crop_clamp_f32(_p(x), C.c_longlong(_bs(x)), Hs, Ws, _p(y), _p(y8), N, Cc, H, W, _stream()), "crop_clamp")
    return (y, y8) if want_u8 else y


# ------------------------------------------------------------------------------------------- VQ
def vq_argmin(z: Tensor, codebook: Tensor, want_zq: bool = True, want_feat: bool = False):
    N, D, H, W = _chk4(z, "vq z")
    if not z.is_contiguous():
        raise ValueError("vq_argmin needs a contiguous latent")
    n_e = codebook.shape[0]
    idx = torch.empty((N, H, W), dtype=torch.int64, device=z.device)
    zq = new(N, D, H, W, z) if want_zq else None
    feat = new(N, D + n_e, H, W, z) if want_feat else None
    check(lib().dcvic_vq_argmin_f32(_p(z), _p(codebook), _p(idx), _p(zq), _p(feat), N, D, H * W, n_e, _stream()), "vq_argmin")
    return idx, zq, feat


def argmax_lut(logits: Tensor, codebook: Tensor, pq_w: Tensor, pq_b: Optional[Tensor]):
    N, n_e, H, W = _chk4(logits, "argmax logits")
    if not logits.is_contiguous():
        raise ValueError("argmax_lut needs contiguous logits")
    D = codebook.shape[1]
    idx = torch.empty((N, H, W), dtype=torch.int64, device=logits.device)
    lat = new(N, D, H, W, logits)
    check(lib().dcvic_argmax_lut_f32(_p(logits), _p(idx), _p(lat), _p(codebook), _p(pq_w), _p(pq_b), N, n_e, D, H * W, _stream()),
          "argmax_lut")
    return idx, lat


# ------------------------------------------------------------------------------------------- rate
def gaussian_rate(y: Optional[Tensor], sym_in: Optional[Tensor], mu: Tensor, sigma: Tensor, scale_table: Tensor,
                  y_hat: Optional[Tensor], sym_out: Optional[Tensor], index_out: Optional[Tensor], lik_out: Optional[Tensor],
                  bits_out: Optional[Tensor]):
    """mu/sigma: [N, C, H, W] views with a common batch stride; sym/index/lik buffers: int32/fp32
    [N, C, H, W] contiguous slices sharing one batch stride."""
    N, Cc, H, W = _chk4(mu, "rate mu")
    _chk4(sigma, "rate sigma")
    if _bs(mu) != _bs(sigma):
        raise ValueError("mu and sigma must share a batch stride")
    si_bs = None
    for t in (sym_in, sym_out, index_out, lik_out):
        if t is not None:
            b = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), Cc * H * W)
            si_bs = b if si_bs is None else si_bs
            if b != si_bs:
                raise ValueError("symbol/index/likelihood buffers must share a batch stride")
    if si_bs is None:
        si_bs = Cc * H * W
    ws = None
    if bits_out is not None:
        ws = torch.empty(N * lib().dcvic_rate_blocks(C.c_longlong(Cc * H * W)), dtype=torch.float64, device=mu.device)
    check(lib().dcvic_gaussian_rate_f32(_p(y), C.c_longlong(_bs(y) if y is not None else 0), _p(sym_in), _p(mu), _p(sigma),
                                        C.c_longlong(_bs(mu)), _p(scale_table), scale_table.numel(), _p(y_hat),
                                        C.c_longlong(_bs(y_hat) if y_hat is not None else 0), _p(sym_out), _p(index_out),
                                        C.c_longlong(si_bs), _p(lik_out), _p(bits_out), _p(ws), N, Cc, H * W, _stream()), "gaussian_rate")


def neglog2_sum(lik: Tensor) -> Tensor:
    """Per-image bit cost -sum(log2 p) of a likelihood map [N, C, H, W] -> float32 [N]."""
    N, Cc, H, W = _chk4(lik, "neglog2 lik")
    out = torch.empty(N, dtype=torch.float32, device=lik.device)
    ws = torch.empty(N * lib().dcvic_rate_blocks(C.c_longlong(Cc * H * W)), dtype=torch.float64, device=lik.device)
    check(lib().dcvic_neglog2_sum_f32(_p(lik), C.c_longlong(_bs(lik)), _p(out), _p(ws), N, C.c_longlong(Cc * H * W), _stream()), "neglog2_sum")
    return out


def eb_rate(z: Optional[Tensor], packs, z_hat: Optional[Tensor], sym_out: Optional[Tensor], lik_out: Optional[Tensor],
            bits_out: Optional[Tensor], sym_in: Optional[Tensor] = None):
    ref = z if z is not None else sym_in
    N, Cc, H, W = ref.shape
    if not ref.is_contiguous():
        raise ValueError("eb_rate needs a contiguous input")
    mats, biases, factors, med = packs
    check(lib().dcvic_eb_rate_f32(_p(z), _p(sym_in), _p(mats), _p(biases), _p(factors), _p(med), _p(z_hat), _p(sym_out), _p(lik_out),
                                  _p(bits_out), N, Cc, H * W, _stream()), "eb_rate")


# ------------------------------------------------------------------------------------------- host entropy coder
def pmf_to_quantized_cdf(pmf: np.ndarray) -> np.ndarray:
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(pmf.size + 1, dtype=np.int32)
    check(lib().dcvic_pmf_to_quantized_cdf_host(pmf.ctypes.data_as(C.c_void_p), int(pmf.size), out.ctypes.data_as(C.c_void_p)),
          "pmf_to_quantized_cdf")
    return out


class CdfTables:
    def __init__(self, cdf: np.ndarray, sizes: np.ndarray, offsets: np.ndarray):
        self.cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        self.sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        h = lib().dcvic_tables_create_host(self.cdf.ctypes.data_as(C.c_void_p), int(self.cdf.shape[0]), int(self.cdf.shape[1]),
                                           self.sizes.ctypes.data_as(C.c_void_p), self.offsets.ctypes.data_as(C.c_void_p))
        if not h:
            check(-1, "tables_create")
        self.h = C.c_void_p(h)

    def __del__(self):
        try:
            lib().dcvic_tables_destroy_host(self.h)
        except Exception:
            pass

    def encode(self, symbols: np.ndarray, indexes: np.ndarray, threads: int = 1) -> List[bytes]:
        """symbols/indexes: int32 [n_streams, n_sym] (host)."""
        symbols = np.ascontiguousarray(symbols, dtype=np.int32)
        indexes = np.ascontiguousarray(indexes, dtype=np.int32)
        ns, n = symbols.shape
        cap = int(n) * 8 + 64
        out = np.empty((ns, cap), dtype=np.uint8)
        lens = np.zeros(ns, dtype=np.int64)
        check(lib().dcvic_rans_encode_batch_host(self.h, symbols.ctypes.data_as(C.c_void_p), indexes.ctypes.data_as(C.c_void_p),
                                                 ns, C.c_longlong(n), out.ctypes.data_as(C.c_void_p), C.c_longlong(cap),
                                                 lens.ctypes.data_as(C.c_void_p), threads), "rans_encode")
        return [out[i, : lens[i]].tobytes() for i in range(ns)]

    def decoders(self, streams: Sequence[bytes]) -> "DecoderSet":
        return DecoderSet(self, streams)


class DecoderSet:
    def __init__(self, tables: CdfTables, streams: Sequence[bytes]):
        self.t = tables
        self.handles = []
        self._bufs = []
        for s in streams:
            buf = np.frombuffer(s, dtype=np.uint8)
            h = lib().dcvic_rans_decoder_create_host(buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.size))
            if not h:
                self.close()
                check(-4, "rans_decoder_create")
            self.handles.append(h)
        self.arr = (C.c_void_p * len(self.handles))(*self.handles)

    def decode(self, indexes: np.ndarray, threads: int = 1) -> np.ndarray:
        indexes = np.ascontiguousarray(indexes, dtype=np.int32)
        ns, n = indexes.shape
        assert ns == len(self.handles)
        out = np.empty((ns, n), dtype=np.int32)
        check(lib().dcvic_rans_decode_batch_host(self.t.h, self.arr, indexes.ctypes.data_as(C.c_void_p), ns, C.c_longlong(n),
                                                 out.ctypes.data_as(C.c_void_p), threads), "rans_decode")
        return out

    def close(self):
        for h in self.handles:
            lib().dcvic_rans_decoder_destroy_host(C.c_void_p(h))
        self.handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
