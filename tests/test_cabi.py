"""The C-ABI library loads and exports every symbol include/dcvic.h declares (no compute calls: CPU box)."""
import os
import re

from dc_vic_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dcvic.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dcvic_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dcvic.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_error_reporting_without_gpu():
    L = _lib.lib()
    assert L.dcvic_version() >= 100
    # argument validation happens before any HIP call
    rc = L.dcvic_pmf_to_quantized_cdf_host(None, 0, None)
    assert rc == -1 and b"pmf_to_quantized_cdf" in L.dcvic_last_error()
    d = _lib.ConvDesc()
    import ctypes as C
    assert L.dcvic_conv_desc_init(C.byref(d), 8, 8, 7, 7, 1, 0, 0, 0) == -1     # 49 taps > 25
    assert L.dcvic_conv_desc_init(C.byref(d), 128, 128, 3, 3, 1, 1, 1, 0) == 0
    assert d.T == 9 and d.tap_dy[0] == -1 and d.tap_dx[8] == 1
    assert L.dcvic_conv_packed_bytes(C.byref(d)) == 128 * 128 * 9 * 4
    assert L.dcvic_convT_phase_desc(C.byref(d), 192, 192, 5, 0, 0) == 0 and d.T == 9
    assert L.dcvic_convT_phase_desc(C.byref(d), 192, 192, 5, 1, 1) == 0 and d.T == 4
    assert L.dcvic_convT_phase_desc(C.byref(d), 192, 192, 5, 0, 1) == 0 and d.T == 6


def test_wino_packed_bytes_host_formula():
    """dcvic_wino_packed_bytes is a host-side size query (no GPU): one 32 KiB slab (16 positions x 8 channels x 64 output
    channels, fp32) per (64-channel tile, 8-channel chunk), both rounded up."""
    import ctypes as C
    from dc_vic_amd import _lib
    L = _lib.lib()
    L.dcvic_wino_packed_bytes.restype = C.c_size_t
    for cin, cout in ((128, 128), (704, 512), (8, 64), (20, 96), (256, 3)):
        assert L.dcvic_wino_packed_bytes(cin, cout) == ((cout + 63) // 64) * ((cin + 7) // 8) * 16 * 8 * 64 * 4
    assert L.dcvic_wino_packed_bytes(0, 64) == 0
    # F(4x4, 3x3): one 72 KiB slab (2 k-steps x 36 positions x 4 channels x 64 output channels) per (64-channel tile, 8-channel chunk)
    L.dcvic_wino44_packed_bytes.restype = C.c_size_t
    for cin, cout in ((256, 256), (704, 512), (8, 64), (128, 200)):
        assert L.dcvic_wino44_packed_bytes(cin, cout) == ((cout + 63) // 64) * ((cin + 7) // 8) * 2 * 36 * 4 * 64 * 4
    assert L.dcvic_wino44_packed_bytes(64, 0) == 0
