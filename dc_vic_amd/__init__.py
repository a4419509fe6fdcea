"""dc_vic_amd -- MI355X-native (gfx950) compress / decompress path of the DC-VIC learned image codec.

Public surface (same names as the reference's `src.models` / `src.utils`):
    build_comp_model(opt), build_subnet(subnet_opt, subnet_type), the *_REGISTRY objects,
    BaseConfig (YAML with _base_/_delete_), HeaderHandler / save_byte_strings / load_byte_strings.
Compute lives in libdcvic_hip.so (dc_vic_amd/csrc, C ABI in include/dcvic.h); importing this package
does not need a GPU, running a model does and fails loudly without the library.
"""
from .registry import *  # noqa: F401,F403
from .options import BaseConfig, ConfigDict  # noqa: F401
from .codec_utils import HeaderHandler, load_byte_strings, save_byte_strings  # noqa: F401


def build_comp_model(opt):
    from .comp_model import build_comp_model as _b
    return _b(opt)


def build_subnet(subnet_opt, subnet_type):
    from .comp_model import build_subnet as _b
    return _b(subnet_opt, subnet_type)


__version__ = "0.1.0"
