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


def test_compute_modes_and_workspace_queries_are_host_only():
    """Mode selection and size queries are pure host state (no launch): the four matrix-core modes round-trip, an
    unknown one is refused, and the attention workspace query includes the backward's dS / P hand-over tiles."""
    import pytest
    from deepsense6g_tii_amd import ops
    L = _lib.lib()
    try:
        for name, code in (("f32", 0), ("bf16", 1), ("f32x3", 2), ("f32x6", 3)):
            ops.set_compute_mode(name)
            assert L.get_compute_mode() == code and ops.get_compute_mode() == name
        with pytest.raises(ValueError):
            ops.set_compute_mode("fp8")
        with pytest.raises(_lib.Ds6gError):
            L.set_compute_mode(4)
        assert ops.get_compute_mode() == "f32x6"   # a refused call leaves the mode alone
    finally:
        ops.set_compute_mode("f32")
    B, T, nh, hd = 12, 962, 4, 128
    C = nh * hd
    slab = B * T * C * 4
    tiles = B * nh * 32 * 32 * 4096          # 8 blocks of 128 keys -> 32 key groups x 32 query tiles of 32 x 32 floats
    need = L.attention_workspace_bytes(B, T, nh, hd, C)
    assert need >= 16 * slab + 2 * tiles       # split slabs + dS and P tiles at hd = 128
    assert L.attention_workspace_bytes(B, T, nh, 64, 256) >= 16 * (slab // 2) + tiles
