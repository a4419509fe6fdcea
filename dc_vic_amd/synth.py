"""Deterministic synthetic weights ("random-init weights of that architecture").

No checkpoint of the reference can be fetched (README.md:38-46 are remote links), so
tests, goldens and bench.py all use weights that are a pure function of
(seed, parameter name, shape).  The generator is independent of module construction
order, of torch's default initialisers and of the device, so the reference modules
(in the build container, oracle/gen_golden.py), the CPU oracle and the HIP product
all see bit-identical tensors.

Scales are chosen so that activations stay O(1) through ~40 layers without any
trained statistics: fan-in scaled normal for conv/linear kernels, (1 + small) for
normalisation gains, small biases, and the reference's own U(-1/n_e, 1/n_e) for the
VQ codebook (taming/modules/vqvae/quantize.py:229-230).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

__all__ = ["synth_tensor", "synth_state_dict"]


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


# per-subnet variance budget: the ELIC nets have no normalisation layers, so they get a
# smaller budget than the GroupNorm/LayerNorm-regularised VQGAN / Swin nets.
_VAR = {"encoder": 1.1, "decoder": 0.8, "hyperencoder": 1.5, "hyperdecoder": 1.6, "context_model": 1.2}


def _gain_for(name: str) -> float:
    # last conv of a residual branch: keep the skip path dominant so depth does not
    # blow the variance up (there are no trained statistics to rely on).
    tails = (
        ".conv.4.weight",      # elic BaseBlock (elic_layers.py:18-24)
        ".c3.weight",          # NLAMResBlock (cheng_nlam.py:36)
        ".conv2.weight",       # ldm ResnetBlock / codeformer ResBlock second conv
        ".conv.5.weight",      # femasr ResBlock second conv (femasr_layers.py:72-79)
        ".proj_out.weight",    # ldm AttnBlock
        ".attn.proj.weight",   # swin attention projection
        ".mlp.fc2.weight",     # swin MLP
    )
    # rate regime: an untrained entropy model would spend ~10 bpp; shrink the latent and the predicted
    # means so that, as in the trained codec (0.05-0.2 bpp), most symbols are zero
    if name == "encoder.conv4.weight":
        return 0.06
    if name.startswith(("encoder.beta_ft_list.7.shift", "encoder.beta_ft_list.8.shift")):
        return 0.05
    if name.endswith(".model.4.weight") and ("mean_slice_transforms" in name or "lrp_slice_transforms" in name):
        return 0.15
    if name.endswith(tails):
        return 0.5
    if ".scale.weight" in name or ".shift.weight" in name:  # BetaScaleShiftModule heads
        return 0.3
    if ".scale.2.weight" in name or ".shift.2.weight" in name:  # SFT heads
        return 0.3
    return 1.0


def synth_tensor(name: str, shape: Tuple[int, ...], dtype: str = "float32", seed: int = 1234) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    g = _rng(seed, name)
    if dtype in ("int64", "int32"):
        raise ValueError(f"integer buffer {name} must be produced by its owner module")
    if name.endswith("quantize.embedding.weight"):
        n_e = shape[0]
        arr = g.uniform(-1.0 / n_e, 1.0 / n_e, size=shape)
    elif leaf == "relative_position_bias_table":
        arr = 0.02 * np.clip(g.standard_normal(shape), -2, 2)
    elif leaf == "weight" and len(shape) >= 2:
        if len(shape) == 4 and _is_conv_transpose(name):
            fan_in = shape[0] * shape[2] * shape[3] / 4.0  # stride-2 transposed conv: ~k*k/4 taps hit
            if shape[2] == 3:
                fan_in = shape[0] * 9
        else:
            fan_in = int(np.prod(shape[1:]))
        std = _gain_for(name) * np.sqrt(_VAR.get(name.split(".", 1)[0], 1.6) / max(fan_in, 1))
        arr = std * g.standard_normal(shape)
    elif leaf == "weight":  # 1-D: normalisation gains
        arr = 1.0 + 0.1 * g.standard_normal(shape)
    elif leaf == "bias":
        arr = 0.05 * g.standard_normal(shape)
    else:
        arr = 0.1 * g.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))


def _is_conv_transpose(name: str) -> bool:
    # ConvTranspose2d layers on the path: hyperdecoder.hd_{mu,std}.conv{1,2,3}
    # (minnen20_hyperprior.py:47-49) and decoder.conv{1,2,3,4} (elic_autoencoder.py:21-28).
    parts = name.split(".")
    if parts[0] == "hyperdecoder":
        return True
    if parts[0] == "decoder" and len(parts) == 3 and parts[1] in ("conv1", "conv2", "conv3", "conv4"):
        return True
    return False


def synth_state_dict(manifest: Dict[str, Iterable], seed: int = 1234) -> Dict[str, torch.Tensor]:
    """manifest: name -> (shape, dtype-string).  Integer buffers are skipped (owners rebuild them)."""
    out = {}
    for name in sorted(manifest):
        shape, dtype = manifest[name]
        if dtype != "float32":
            continue
        if name.endswith(".attn_mask"):
            continue
        out[name] = synth_tensor(name, tuple(shape), dtype, seed)
    return out


# ------------------------------------------------------------------------------------------
# Full-model manifest: the reference-importable sub-modules come from the committed manifest
# (tests/golden/state_dict_manifest.json is produced from the reference's own modules); the
# CompressAI-dependent ones are derived from their source:
#   context_model.*   minnen20_charm_context_model.py:53-68
#   entropy_model_z.* CompressAI 1.2.4 EntropyBottleneck(192) parameter names (App-B)
def charm_manifest(num_slices: int = 6, slice_ch: int = 32, hyper_half: int = 128, max_support: int = 4):
    man = {}
    for i in range(num_slices):
        sup = slice_ch * min(i, max_support)
        for kind, cin in (("mean", sup + hyper_half), ("scale", sup + hyper_half), ("lrp", sup + hyper_half + slice_ch)):
            p = f"context_model.{kind}_slice_transforms.{i}.model"
            man[p + ".0.weight"] = [[224, cin, 5, 5], "float32"]; man[p + ".0.bias"] = [[224], "float32"]
            man[p + ".2.weight"] = [[128, 224, 5, 5], "float32"]; man[p + ".2.bias"] = [[128], "float32"]
            man[p + ".4.weight"] = [[slice_ch, 128, 3, 3], "float32"]; man[p + ".4.bias"] = [[slice_ch], "float32"]
    return man


def _eb_params(channels: int, seed: int, prefix: str) -> Dict[str, torch.Tensor]:
    """EntropyBottleneck parameters: the library's initial values perturbed per channel so that
    CDF tables differ in length/offset (exercises ragged tables)."""
    g = np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, 0xEB]))
    filters = (1, 3, 3, 3, 3, 1)
    scale = 0.6 ** (1 / 5)      # a sharper density than the library's init_scale=10: trained z priors are narrow
    out = {}
    for i in range(5):
        init = np.log(np.expm1(1 / scale / filters[i + 1]))
        out[f"{prefix}._matrix{i}"] = torch.from_numpy((init + 0.1 * g.standard_normal((channels, filters[i + 1], filters[i]))).astype(np.float32))
        out[f"{prefix}._bias{i}"] = torch.from_numpy(g.uniform(-0.5, 0.5, (channels, filters[i + 1], 1)).astype(np.float32))
        if i < 4:
            out[f"{prefix}._factor{i}"] = torch.from_numpy((0.2 * g.standard_normal((channels, filters[i + 1], 1))).astype(np.float32))
    med = 0.3 * g.standard_normal(channels)
    lo = med - g.uniform(2.0, 9.0, channels)
    hi = med + g.uniform(2.0, 9.0, channels)
    out[f"{prefix}.quantiles"] = torch.from_numpy(np.stack([lo, med, hi], axis=1)[:, None, :].astype(np.float32))
    return out


def reference_manifest() -> Dict[str, list]:
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "manifest", "state_dict_manifest.json")
    with open(path) as f:
        return json.load(f)


def full_synth_state_dict(seed: int = 1234) -> Dict[str, torch.Tensor]:
    man = dict(reference_manifest())
    man.update(charm_manifest())
    sd = synth_state_dict(man, seed)
    sd.update(_eb_params(192, seed, "entropy_model_z"))
    return sd


def load_synth_weights(model, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Load the deterministic synthetic weights into a dc_vic_amd comp model (in place) and return
    the CPU state dict that was loaded (the oracle consumes the same dict)."""
    sd = full_synth_state_dict(seed)
    own = model.state_dict()
    missing = [k for k in sd if k not in own]
    if missing:
        raise KeyError(f"synthetic weights name parameters the model lacks: {missing[:5]}")
    merged = {k: (sd[k] if k in sd else v) for k, v in own.items()}
    model.load_state_dict(merged)
    return sd


def synth_discriminator_state(shapes: Dict[str, Iterable], seed: int = 5) -> Dict[str, torch.Tensor]:
    """Deterministic PatchGAN weights for fixtures and tests: name -> shape in, name -> tensor out, drawn in SORTED key order from one
    CPU generator (so the reference module in oracle/gen_golden.py and the product module get bit-identical values without the
    11 MB state dict being committed).  N(0, 0.05) matrices / kernels, N(0, 0.02) biases -- a bit larger than taming's
    weights_init N(0, 0.02) so the logits carry signal."""
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for name in sorted(shapes):
        shp = tuple(int(v) for v in shapes[name])
        out[name] = torch.randn(shp, generator=g) * (0.05 if len(shp) > 1 else 0.02)
    return out
