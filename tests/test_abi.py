"""CPU-only: the C-ABI library builds/loads and exports every symbol include/asw_hip.h
declares; the ctypes table covers exactly that set.  No compute call is made."""
import os
import re

from acousticswarms_speech_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "asw_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(asw_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    native.build()
    L = native.lib()
    syms = _declared()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in asw_hip.h but not exported"
    assert syms == set(native.SIGNATURES), (syms ^ set(native.SIGNATURES))
    assert L.asw_abi_version() == 3


def test_argument_errors_do_not_need_a_gpu():
    """Bad arguments are rejected before any HIP call (negative status + message)."""
    from ctypes import byref, c_void_p
    L = native.lib()
    cfg = native.SpotConfigC()
    h = c_void_p()
    assert L.asw_spot_create(byref(cfg), byref(h)) == -1          # depth 0
    assert b"depth" in L.asw_last_error()
    assert L.asw_convgemm_f32(None, None) == -1
    assert L.asw_spot_shift_and_sep(None, None, 7, 100, None, 1, 0, 1, None, None, 0, None) == -1
    assert L.asw_segment_sisdr(None, 4, 100, None, None, 2, None, None) == -1
    assert b"segment_sisdr" in L.asw_last_error()
    assert L.asw_add_layernorm(None, None, None, None, 4, 1024, 1e-5, None, None) == -1


def test_host_entry_points_run_without_a_gpu():
    """asw_cube_select / asw_search_area are plain host functions of the library."""
    import numpy as np
    from ctypes import byref, c_int64, c_void_p
    L = native.lib()
    rng = np.random.default_rng(0)
    off = np.ascontiguousarray(rng.uniform(-10, 10, (5, 7, 3, 4)))
    c, w = np.array([1.0, -2.0, 0.5, 3.0]), 9.0
    lo, hi = np.ascontiguousarray(c - w / 2), np.ascontiguousarray(c + w / 2)
    idx = np.empty(5 * 7 * 3, dtype=np.int32)
    n = c_int64()
    native.check(L.asw_cube_select(c_void_p(off.ctypes.data), 5, 7, 3, 4, 1, 4, 2, 6, c_void_p(lo.ctypes.data),
                                   c_void_p(hi.ctypes.data), c_void_p(idx.ctypes.data), idx.size, byref(n)))
    mask = np.zeros((5, 7, 3), dtype=bool)
    mask[1:4, 2:6] = np.all((off[1:4, 2:6] >= lo) & (off[1:4, 2:6] <= hi), axis=-1)
    np.testing.assert_array_equal(idx[:n.value], np.flatnonzero(mask))
    # the pair-major variant: same hits, same order (ragged run lengths exercise the blocked first-pair test)
    planes = np.ascontiguousarray(np.moveaxis(off, 3, 0))
    idx2 = np.empty_like(idx)
    for box in [(1, 4, 2, 6), (0, 5, 0, 7), (2, 3, 1, 2)]:
        y0, y1, x0, x1 = box
        native.check(L.asw_cube_select(c_void_p(off.ctypes.data), 5, 7, 3, 4, y0, y1, x0, x1, c_void_p(lo.ctypes.data),
                                       c_void_p(hi.ctypes.data), c_void_p(idx.ctypes.data), idx.size, byref(n)))
        n1 = n.value
        native.check(L.asw_cube_select_planes(c_void_p(planes.ctypes.data), 5, 7, 3, 4, y0, y1, x0, x1,
                                              c_void_p(lo.ctypes.data), c_void_p(hi.ctypes.data), c_void_p(idx2.ctypes.data),
                                              idx2.size, byref(n)))
        assert n.value == n1 and (n1 > 0 or box == (2, 3, 1, 2))
        np.testing.assert_array_equal(idx2[:n1], idx[:n1])
    # capacity too small -> status + message, no overrun
    assert L.asw_cube_select(c_void_p(off.ctypes.data), 5, 7, 3, 4, 0, 5, 0, 7, c_void_p((c - 100).ctypes.data),
                             c_void_p((c + 100).ctypes.data), c_void_p(idx.ctypes.data), 3, byref(n)) == -1
    assert b"capacity" in L.asw_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        native.lib()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected RuntimeError")


def test_header_is_plain_c():
    """include/asw_hip.h must be consumable from C (cgo / JNI / ctypes-style bindings)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror",
                    os.path.join(root, "include", "asw_hip.h")], check=True)


def test_torch_custom_ops_are_registered_without_a_gpu():
    """TORCH_LIBRARY(asw, ...): the adapter library builds, loads and registers every op of
    SURVEY.md §8(b); tensors that are not on the GPU are rejected by the dispatcher (the ops have
    a CUDA/HIP implementation only -- there is no CPU fallback to fall into)."""
    import pytest
    import torch
    native.build_torch_ops()
    ops = native.torch_ops()
    for name in ("spot_shift_and_sep", "spot_forward", "shift_norm_preproc", "energies", "pair_sisdr", "segment_sisdr",
                 "center_rows_", "srp_phat_map", "sep_infer", "sep_forward"):
        assert hasattr(ops, name), name
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.pair_sisdr(torch.zeros(2, 8))
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.energies(torch.zeros(2, 8), 4)


def test_sep_argument_errors_do_not_need_a_gpu():
    from ctypes import byref, c_void_p
    L = native.lib()
    cfg = native.SepConfigC()
    h = c_void_p()
    assert L.asw_sep_create(byref(cfg), byref(h)) == -1           # depth 0
    assert b"depth" in L.asw_last_error()
    assert L.asw_sep_infer(None, None, 7, 100, None, 2, None, None) == -1
    assert L.asw_relpos_attention(None, None, None, None, 1, 10, 128, 8, 0.1, None, None) == -1
    assert L.asw_add_layernorm2(None, None, 1.0, None, None, 4, 512, 1e-5, 0, None, None, None) == -1
    assert L.asw_joint_shift_stats_scratch_doubles() > 0
