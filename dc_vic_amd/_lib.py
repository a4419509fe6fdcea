"""ctypes binding of libdcvic_hip.so (include/dcvic.h).  The library is REQUIRED: there is no
fallback path -- if it is missing or an entry point fails, the caller gets an exception."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCVIC_LIB_PATH") or os.path.join(_HERE, "libdcvic_hip.so")   # DCVIC_LIB_PATH: diagnostic builds (tools/)

MAX_TAPS = 25
MAX_SRC = 3

ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SWISH, ACT_GELU, ACT_SIGMOID, ACT_HALF_TANH = range(7)


class ConvDesc(C.Structure):
    _fields_ = [
        ("Cin", C.c_int), ("Cout", C.c_int), ("T", C.c_int),
        ("tap_ky", C.c_int8 * MAX_TAPS), ("tap_kx", C.c_int8 * MAX_TAPS),
        ("tap_dy", C.c_int8 * MAX_TAPS), ("tap_dx", C.c_int8 * MAX_TAPS),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("upsample", C.c_int),
        ("transposed_weight", C.c_int), ("cfg", C.c_int),
    ]


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int), ("batch_stride", C.c_longlong)]


class ConvIO(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("Hout", C.c_int), ("Wout", C.c_int), ("Hfull", C.c_int), ("Wfull", C.c_int),
        ("osy", C.c_int), ("osx", C.c_int), ("ooy", C.c_int), ("oox", C.c_int),
        ("n_src", C.c_int), ("src", Src * MAX_SRC),
        ("out", C.c_void_p), ("out_batch_stride", C.c_longlong),
        ("bias", C.c_void_p), ("act", C.c_int),
        ("res", C.c_void_p), ("res_batch_stride", C.c_longlong),
        ("aff_scale", C.c_void_p), ("aff_shift", C.c_void_p), ("aff_batch_stride", C.c_longlong),
        ("init", C.c_void_p), ("init_batch_stride", C.c_longlong),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("a_bs", C.c_longlong), ("a_ms", C.c_longlong), ("a_ks", C.c_longlong),
        ("B", C.c_void_p), ("b_bs", C.c_longlong), ("b_ks", C.c_longlong), ("b_ns", C.c_longlong),
        ("C", C.c_void_p), ("c_bs", C.c_longlong), ("c_ms", C.c_longlong),
        ("alpha", C.c_float),
    ]


# every symbol include/dcvic.h declares (tests/test_cabi.py checks the library exports all of them)
SYMBOLS = [
    "dcvic_last_error", "dcvic_version", "dcvic_device_info",
    "dcvic_conv_desc_init", "dcvic_convT_phase_desc", "dcvic_conv_select_class", "dcvic_conv_last_variant", "dcvic_conv_set_tuning", "dcvic_conv_packed_bytes", "dcvic_conv_pack_f32", "dcvic_conv2d_f32",
    "dcvic_wino_packed_bytes", "dcvic_wino_pack_f32", "dcvic_conv3x3_wino_f32",
    "dcvic_wino_ups_packed_bytes", "dcvic_wino_ups_pack_f32", "dcvic_conv3x3_wino_ups_f32",
    "dcvic_wino44_packed_bytes", "dcvic_wino44_pack_f32", "dcvic_conv3x3_wino44_f32",
    "dcvic_conv3x3_thin_applies", "dcvic_conv3x3_thin_f32",
    "dcvic_wino44_stats_tiles", "dcvic_conv3x3_wino44_stats_f32", "dcvic_groupnorm_part_f32",
    "dcvic_bgemm_f32", "dcvic_attn_fused_f32", "dcvic_groupnorm_f32", "dcvic_layernorm_c_f32", "dcvic_softmax_c_f32", "dcvic_swin_attn_f32",
    "dcvic_ew_f32", "dcvic_chan_affine_f32", "dcvic_copy_planes_f32", "dcvic_copy_window_f32", "dcvic_absmax_f32", "dcvic_crop_clamp_f32",
    "dcvic_vq_argmin_f32", "dcvic_argmax_lut_f32", "dcvic_gaussian_rate_f32", "dcvic_rate_blocks", "dcvic_neglog2_sum_f32", "dcvic_eb_rate_f32",
    "dcvic_pmf_to_quantized_cdf_host", "dcvic_tables_create_host", "dcvic_tables_destroy_host",
    "dcvic_rans_encode_batch_host", "dcvic_rans_decoder_create_host", "dcvic_rans_decoder_destroy_host",
    "dcvic_rans_decode_batch_host",
    # training step (csrc/train.hip)
    "dcvic_conv_wgrad_workspace_floats", "dcvic_conv_wgrad_f32", "dcvic_chan_reduce_f32", "dcvic_sum_rows_f32", "dcvic_ew_bwd_f32",
    "dcvic_groupnorm_bwd_f32", "dcvic_layernorm_c_bwd_blocks", "dcvic_layernorm_c_bwd_f32", "dcvic_softmax_c_bwd_f32",
    "dcvic_swin_attn_bwd_f32", "dcvic_reduce_loss_f32", "dcvic_cross_entropy_f32", "dcvic_adam_step_f32", "dcvic_clip_scale_f32",
    "dcvic_resample2_f32", "dcvic_s2d_f32", "dcvic_maxpool3s2_f32", "dcvic_lpips_tap_f32",
]

_lib = None


class DcvicError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library; raise loudly if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DcvicError(
            f"{LIB_PATH} is missing: build it with `python dc_vic_amd/csrc/build.py` "
            "(or __graft_entry__.build()). dc_vic_amd has no fallback path.")
    L = C.CDLL(LIB_PATH)
    L.dcvic_last_error.restype = C.c_char_p
    L.dcvic_conv_packed_bytes.restype = C.c_size_t
    L.dcvic_wino_packed_bytes.restype = C.c_size_t
    L.dcvic_wino_ups_packed_bytes.restype = C.c_size_t
    L.dcvic_wino44_packed_bytes.restype = C.c_size_t
    L.dcvic_conv_wgrad_workspace_floats.restype = C.c_longlong
    L.dcvic_tables_create_host.restype = C.c_void_p
    L.dcvic_rans_decoder_create_host.restype = C.c_void_p
    L.dcvic_tables_destroy_host.argtypes = [C.c_void_p]
    L.dcvic_rans_decoder_destroy_host.argtypes = [C.c_void_p]
    L.dcvic_conv_packed_bytes.argtypes = [C.POINTER(ConvDesc)]
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().dcvic_last_error()
        raise DcvicError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
