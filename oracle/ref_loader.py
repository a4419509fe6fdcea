"""Loader for the *reference's own* torch modules (container-only tooling).

TEST INFRASTRUCTURE, NOT PRODUCT.  Only oracle/gen_golden.py imports this, and
only in the build container where /root/reference exists; nothing here travels
to the GPU box and the product never imports it.

SURVEY.md section 8c: `import src.models` fails because the package __init__
files glob-import every sub-module (first failure python_log_indenter, then
compressai).  We therefore pre-seed sys.modules with *path-only* package
objects for the `src.*` packages (so their __init__.py never executes) and
provide three NON-ARITHMETIC helper stand-ins:
  * python_log_indenter.IndentedLoggerAdapter  (logging only)
  * timm.models.layers.{to_2tuple, trunc_normal_, DropPath}
      - to_2tuple: pure Python tuple helper
      - trunc_normal_: weight *initialiser* (all weights are overwritten by
        dc_vic_amd.synth weights before any golden is taken)
      - DropPath: never instantiated (drop_path=0. -> nn.Identity in
        swinir_layers.py:193)
None of them participates in a forward pass.  CompressAI-dependent files
(src/models/comp_model/*, entropy_model/*) are NOT imported: they stay "parity
unpinned" (SURVEY 8c).

`install_compressai_names()` (used only for the CHARM fixture) additionally
registers NAME-ONLY placeholders for the two identifiers that
minnen20_charm_context_model.py:12-13 imports at module level:
  * compressai.ans.RansDecoder            - only instantiated in forward_decompress (:179)
  * compressai.entropy_models.GaussianConditional - only a type annotation (:74, :124, :173)
Both placeholders raise if anything tries to use them, so no CompressAI
arithmetic is faked: the fixture drives `forward()` only, whose conv / cat /
tanh arithmetic is plain torch, with the entropy-model callable passed in by
gen_golden.py and stated in the fixture's docstring.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("DCVIC_REFERENCE", "/root/reference")


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # helper stand-ins (non-arithmetic)
    pli = types.ModuleType("python_log_indenter")

    class IndentedLoggerAdapter:  # logging only
        def __init__(self, logger, *a, **k):
            self.logger = logger

        def add(self, *a, **k):
            return self

        def sub(self, *a, **k):
            return self

        def __getattr__(self, item):
            return getattr(self.logger, item)

    pli.IndentedLoggerAdapter = IndentedLoggerAdapter
    sys.modules.setdefault("python_log_indenter", pli)

    import itertools
    import collections.abc
    import torch.nn as nn

    def to_2tuple(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return tuple(x)
        return tuple(itertools.repeat(x, 2))

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.to_2tuple = to_2tuple
    timm_layers.trunc_normal_ = lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: nn.init.trunc_normal_(t, mean, std, a, b)

    class DropPath(nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0, "stochastic depth is never enabled on the inference path"

    timm_layers.DropPath = DropPath
    timm.models = timm_models
    timm_models.layers = timm_layers
    sys.modules.setdefault("timm", timm)
    sys.modules.setdefault("timm.models", timm_models)
    sys.modules.setdefault("timm.models.layers", timm_layers)

    # path-only packages so the glob-importing __init__.py files never run
    for name in [
        "src",
        "src.models",
        "src.models.subnet",
        "src.models.subnet.autoencoder",
        "src.models.subnet.hyperprior",
        "src.models.subnet.context_model",
        "src.models.subnet.vq_estimator",
        "src.models.layer",
        "src.utils",
    ]:
        if name not in sys.modules:
            _pkg(name, os.path.join(REF, *name.split(".")))


def install_compressai_names():
    """Name-only placeholders for `from compressai.ans import RansDecoder` and `from compressai.entropy_models
    import GaussianConditional` (minnen20_charm_context_model.py:12-13).  Using either raises."""
    if "compressai" in sys.modules:
        return

    class _NameOnly:
        def __init__(self, *a, **k):
            raise RuntimeError("compressai is not installed: this is a name-only placeholder (oracle/ref_loader.py)")

    ca = types.ModuleType("compressai")
    ans = types.ModuleType("compressai.ans")
    em = types.ModuleType("compressai.entropy_models")
    ans.RansDecoder = type("RansDecoder", (_NameOnly,), {})
    em.GaussianConditional = type("GaussianConditional", (_NameOnly,), {})
    ca.ans, ca.entropy_models = ans, em
    ca.__dcvic_name_only__ = True
    sys.modules["compressai"] = ca
    sys.modules["compressai.ans"] = ans
    sys.modules["compressai.entropy_models"] = em


def install_training_names():
    """For the a20 fixtures (tests/golden/train.npz): path-only packages for `src.models.discriminator` and `src.losses` (their
    __init__.py glob-import every sibling, among them files that need lpips / pytorch_msssim) and ONE name-only placeholder:
      * pytorch_msssim.MS_SSIM -- imported at module level by src/losses/distortion_loss.py:6, used only by the MS-SSIM loss
        class of that file (never instantiated here: the fixture drives MSELoss and VanillaMSELoss).  It raises when used.
    The discriminator and loss classes themselves are the reference's own code, imported from /root/reference."""
    install()
    for name in ["src.models.discriminator", "src.losses"]:
        if name not in sys.modules:
            _pkg(name, os.path.join(REF, *name.split(".")))
    if "pytorch_msssim" not in sys.modules:
        class _NameOnly:
            def __init__(self, *a, **k):
                raise RuntimeError("pytorch_msssim is not installed: this is a name-only placeholder (oracle/ref_loader.py)")
        pm = types.ModuleType("pytorch_msssim")
        pm.MS_SSIM = type("MS_SSIM", (_NameOnly,), {})
        pm.__dcvic_name_only__ = True
        sys.modules["pytorch_msssim"] = pm


def ref(modname):
    install()
    return importlib.import_module(modname)
