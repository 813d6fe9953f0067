"""CPU-side checks of the C-ABI boundary: libds6g.so loads and exports every symbol include/ds6g.h
declares (no compute call is made without a GPU)."""
import ctypes
import os

from deepsense6g_tii_amd import _lib


def test_header_parses():
    protos = _lib.parse_header()
    assert len(protos) >= 35
    for name in ("ds6g_conv2d_fwd", "ds6g_attention_bwd", "ds6g_adamw_step", "ds6g_focal_loss", "ds6g_bn_bwd"):
        assert name in protos
    # pointers are void*, sizes are size_t / long, seeds are 64-bit
    _, args = protos["ds6g_linear_fwd"]
    assert args[0] is ctypes.c_void_p and args[-1] is ctypes.c_void_p
    assert args[-3] is ctypes.c_uint64 and args[-2] is ctypes.c_uint64


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with make -C deepsense6g_tii_amd/csrc"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(dll, name), name
    L = _lib.lib()
    assert L.version() >= 1


def test_argument_errors_are_reported_not_launched():
    # NULL pointers are rejected on the host before any launch (safe without a GPU)
    import pytest
    L = _lib.lib()
    with pytest.raises(_lib.Ds6gError):
        L.conv2d_fwd(0, 0, 0, 1, 8, 8, 4, 4, 3, 3, 1, 1, 0)
    with pytest.raises(_lib.Ds6gError):
        L.layernorm_fwd(0, 0, 0, 0, 0, 0, 4, 64, 1e-5, 0)
