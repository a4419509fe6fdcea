"""Oracle restatement of the CompressAI 1.2.4 entropy models the reference wraps.

TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (CompressAI source is absent from the
reference tree and cannot be installed; no golden vector exists for this boundary).  Restated
from SURVEY.md Appendix B; reference wrappers/call sites:
  src/models/subnet/entropy_model/entropy_bottleneck.py:19-28   (SteEntropyBottleneck)
  src/models/subnet/entropy_model/ste_gaussian_conditional.py:16-23, gaussian_conditional.py:22-24
  src/models/comp_model/hyperprior_dc_vic_model.py:65-68 (codec_setup: update / update_scale_table)
All arithmetic is torch CPU fp32, in the operator order App-B lists.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 256.0, 64   # compressai.models.get_scale_table defaults
SCALE_BOUND = 0.11                                     # ...vq_f8_n256.yaml:58
LIKELIHOOD_BOUND = 1e-9
TAIL_MASS = 1e-9


def build_lib(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "librans_oracle.so")
    src = os.path.join(_HERE, "rans_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build_lib())
        L.oracle_rans_encode.restype = ctypes.c_long
        L.oracle_rans_dec_new.restype = ctypes.c_void_p
        L.oracle_rans_dec_decode.restype = ctypes.c_int
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


# ------------------------------------------------------------------------------- CDF tables
def pmf_to_quantized_cdf(pmf: np.ndarray) -> np.ndarray:
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(pmf.size + 1, dtype=np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(_p(pmf, ctypes.c_float), ctypes.c_int(pmf.size), _p(out, ctypes.c_uint32))
    if rc != 0:
        raise ValueError(f"pmf_to_quantized_cdf failed rc={rc}")
    return out.astype(np.int32)


def pmf_to_cdf(pmf: torch.Tensor, tail_mass: torch.Tensor, pmf_length: torch.Tensor, max_length: int) -> np.ndarray:
    """EntropyModel._pmf_to_cdf (App-B): per row cat(pmf[:len], tail) -> quantised cdf, zero padded."""
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = torch.cat((pmf[i, : int(pmf_length[i])], tail_mass[i].reshape(-1)), dim=0).numpy()
        c = pmf_to_quantized_cdf(prob)
        cdf[i, : c.size] = c
    return cdf


def get_scale_table() -> torch.Tensor:
    return torch.exp(torch.linspace(math.log(SCALE_MIN), math.log(SCALE_MAX), SCALE_LEVELS))


def _std_cumulative(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * torch.erfc(float(-(2 ** -0.5)) * x)


def gc_likelihood(y_hat: torch.Tensor, sigma: torch.Tensor, mu: torch.Tensor) -> torch.Tensor:
    """GaussianConditional._likelihood + likelihood lower bound (App-B)."""
    v = torch.abs(y_hat - mu)
    s = torch.clamp(sigma, min=SCALE_BOUND)       # LowerBound forward = max(x, bound)
    up = _std_cumulative((0.5 - v) / s)
    lo = _std_cumulative((-0.5 - v) / s)
    return torch.clamp(up - lo, min=LIKELIHOOD_BOUND)


def gc_build_indexes(sigma: torch.Tensor) -> torch.Tensor:
    """GaussianConditional.build_indexes (App-B)."""
    table = get_scale_table()
    s = torch.clamp(sigma, min=SCALE_BOUND)
    idx = s.new_full(s.size(), len(table) - 1).int()
    for t in table[:-1]:
        idx -= (s <= t).int()
    return idx


class _Tables:
    cdf: np.ndarray
    cdf_length: np.ndarray
    offset: np.ndarray

    def encode(self, symbols: torch.Tensor, indexes: torch.Tensor) -> bytes:
        s = np.ascontiguousarray(symbols.reshape(-1).numpy(), dtype=np.int32)
        ix = np.ascontiguousarray(indexes.reshape(-1).numpy(), dtype=np.int32)
        cap = s.size * 8 + 64
        out = np.zeros(cap, dtype=np.uint8)
        n = lib().oracle_rans_encode(_p(s, ctypes.c_int32), _p(ix, ctypes.c_int32), ctypes.c_long(s.size),
                                     _p(self.cdf, ctypes.c_int32), ctypes.c_int(self.cdf.shape[1]),
                                     _p(self.cdf_length, ctypes.c_int32), _p(self.offset, ctypes.c_int32),
                                     _p(out, ctypes.c_uint8), ctypes.c_long(cap))
        if n < 0:
            raise RuntimeError(f"oracle_rans_encode rc={n}")
        return out[:n].tobytes()

    class _Dec:
        def __init__(self, tables, stream: bytes):
            self.t = tables
            self.buf = np.frombuffer(stream, dtype=np.uint8).copy()
            self.h = ctypes.c_void_p(lib().oracle_rans_dec_new(_p(self.buf, ctypes.c_uint8), ctypes.c_long(self.buf.size)))

        def decode(self, indexes: torch.Tensor) -> torch.Tensor:
            ix = np.ascontiguousarray(indexes.reshape(-1).numpy(), dtype=np.int32)
            out = np.zeros(ix.size, dtype=np.int32)
            rc = lib().oracle_rans_dec_decode(self.h, _p(ix, ctypes.c_int32), ctypes.c_long(ix.size),
                                              _p(self.t.cdf, ctypes.c_int32), ctypes.c_int(self.t.cdf.shape[1]),
                                              _p(self.t.cdf_length, ctypes.c_int32), _p(self.t.offset, ctypes.c_int32),
                                              _p(out, ctypes.c_int32))
            if rc != 0:
                raise RuntimeError(f"oracle_rans_dec_decode rc={rc}")
            return torch.from_numpy(out).reshape(indexes.shape)

        def __del__(self):
            try:
                lib().oracle_rans_dec_free(self.h)
            except Exception:
                pass

    def stream_decoder(self, stream: bytes):
        return _Tables._Dec(self, stream)


class GaussianConditionalOracle(_Tables):
    """GaussianConditional.update() tables for scale_table = get_scale_table() (App-B)."""

    def __init__(self):
        from scipy.stats import norm
        table = get_scale_table()
        multiplier = -norm.ppf(TAIL_MASS / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(pmf_length.max())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        sc = table.unsqueeze(1).float()
        upper = _std_cumulative((0.5 - samples) / sc)
        lower = _std_cumulative((-0.5 - samples) / sc)
        pmf = upper - lower
        tail = 2 * lower[:, :1]
        self.cdf = pmf_to_cdf(pmf, tail, pmf_length, max_length)
        self.offset = np.ascontiguousarray((-pmf_center).numpy(), dtype=np.int32)
        self.cdf_length = np.ascontiguousarray((pmf_length + 2).numpy(), dtype=np.int32)
        self.scale_table = table

    def compress(self, symbols: torch.Tensor, indexes: torch.Tensor) -> bytes:
        """EntropyModel.compress for one batch item (N must be 1)."""
        assert symbols.shape[0] == 1
        return self.encode(symbols[0], indexes[0])


class EntropyBottleneckOracle(_Tables):
    """EntropyBottleneck (filters (3,3,3,3), tail_mass 1e-9) eval forward / update / compress (App-B)."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str):
        self.p = {k[len(prefix) + 1:]: v.detach().cpu().float() for k, v in sd.items() if k.startswith(prefix + ".")}
        self.C = self.p["quantiles"].shape[0]
        self._update()

    def _logits_cumulative(self, v: torch.Tensor) -> torch.Tensor:
        logits = v
        for i in range(5):
            logits = torch.matmul(F.softplus(self.p[f"_matrix{i}"]), logits)
            logits = logits + self.p[f"_bias{i}"]
            if i < 4:
                logits = logits + torch.tanh(self.p[f"_factor{i}"]) * torch.tanh(logits)
        return logits

    def _likelihood(self, x: torch.Tensor) -> torch.Tensor:
        lower = self._logits_cumulative(x - 0.5)
        upper = self._logits_cumulative(x + 0.5)
        sign = -torch.sign(lower + upper)
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))

    def medians(self) -> torch.Tensor:
        return self.p["quantiles"][:, 0, 1]

    def forward(self, z: torch.Tensor):
        """eval forward: z_hat = round(z - med) + med; likelihood floored at 1e-9."""
        N, C, H, W = z.shape
        v = z.permute(1, 0, 2, 3).contiguous().reshape(C, 1, -1)
        med = self.p["quantiles"][:, :, 1:2]
        out = torch.round(v - med) + med
        lik = torch.clamp(self._likelihood(out), min=LIKELIHOOD_BOUND)
        back = lambda t: t.reshape(C, N, H, W).permute(1, 0, 2, 3).contiguous()
        return back(out), back(lik)

    def symbols(self, z: torch.Tensor) -> torch.Tensor:
        med = self.medians().view(1, -1, 1, 1)
        return torch.round(z - med).int()

    def _update(self):
        q = self.p["quantiles"]
        med = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
        self.offset = np.ascontiguousarray((-minima).numpy(), dtype=np.int32)
        pmf_start = med - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative(samples - 0.5)
        upper = self._logits_cumulative(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self.cdf = pmf_to_cdf(pmf, tail, pmf_length, max_length)
        self.cdf_length = np.ascontiguousarray((pmf_length + 2).numpy(), dtype=np.int32)

    def compress(self, z_symbols: torch.Tensor) -> bytes:
        assert z_symbols.shape[0] == 1
        N, C, H, W = z_symbols.shape
        idx = torch.arange(C, dtype=torch.int32).view(1, C, 1, 1).expand(N, C, H, W)
        return self.encode(z_symbols[0], idx[0])

    def decompress(self, stream: bytes, zH: int, zW: int) -> torch.Tensor:
        idx = torch.arange(self.C, dtype=torch.int32).view(self.C, 1, 1).expand(self.C, zH, zW).contiguous()
        sym = self.stream_decoder(stream).decode(idx)
        return (sym.float() + self.medians().view(-1, 1, 1)).unsqueeze(0)


def synth_entropy_bottleneck(channels: int, seed: int = 1234, prefix: str = "entropy_model_z") -> Dict[str, torch.Tensor]:
    """Deterministic EntropyBottleneck parameters (CompressAI 1.2.4 key names):
    the library's own initial values (matrix init log(expm1(1/scale/f)), factors 0, quantiles
    [-10,0,10]) perturbed so channels differ; quantiles chosen so tables have distinct lengths."""
    g = np.random.Generator(np.random.PCG64([seed, 0xEB]))
    filters = (1, 3, 3, 3, 3, 1)
    scale = 10 ** (1 / 5)
    out = {}
    for i in range(5):
        init = np.log(np.expm1(1 / scale / filters[i + 1]))
        out[f"{prefix}._matrix{i}"] = torch.from_numpy((init + 0.1 * g.standard_normal((channels, filters[i + 1], filters[i]))).astype(np.float32))
        out[f"{prefix}._bias{i}"] = torch.from_numpy(g.uniform(-0.5, 0.5, (channels, filters[i + 1], 1)).astype(np.float32))
        if i < 4:
            out[f"{prefix}._factor{i}"] = torch.from_numpy((0.2 * g.standard_normal((channels, filters[i + 1], 1))).astype(np.float32))
    med = 0.3 * g.standard_normal(channels)
    lo = med - g.uniform(4.0, 12.0, channels)
    hi = med + g.uniform(4.0, 12.0, channels)
    out[f"{prefix}.quantiles"] = torch.from_numpy(np.stack([lo, med, hi], axis=1)[:, None, :].astype(np.float32))
    return out
