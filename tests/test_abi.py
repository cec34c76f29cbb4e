"""The C-ABI boundary: every function include/manta_hip.h declares is exported by the HIP product library and by
the oracle; the binding is generated from the header; the product path refuses to run without its native library."""
import ctypes
import os

import pytest

import util
from mantaflow_amd import _lib


def test_header_parses_and_declares_the_hot_path():
    protos = _lib.parse_header()
    for name in ("mf_apply_matrix", "mf_cg_solve", "mf_mic_apply", "mf_semi_lagrange_mac", "mf_maccormack_clamp",
                 "mf_map_parts_to_mac", "mf_flip_velocity_update", "mf_advect_in_grid", "mf_make_rhs", "mf_correct_velocity"):
        assert name in protos
    assert protos["mf_apply_matrix"][1][:3] == [ctypes.c_int] * 3
    assert protos["mf_grid_dot"][1][0] is ctypes.c_int64


@pytest.mark.parametrize("path", [util.HIP_LIB, util.ORACLE_LIB])
def test_library_exports_every_declared_symbol(path):
    if path == util.ORACLE_LIB:
        util.build_oracle()
    assert os.path.exists(path), "%s missing -- run __graft_entry__.build()" % path
    L = ctypes.CDLL(path)           # loads without a GPU; no compute call is made here
    for name in _lib.parse_header():
        assert hasattr(L, name), "%s lacks %s" % (path, name)
    L.mf_backend.restype = ctypes.c_char_p
    assert L.mf_backend().decode() == ("hip" if path == util.HIP_LIB else "oracle")


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    _lib.reset()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU visible"):
            _lib.get()
    monkeypatch.setattr(_lib, "DEFAULT_LIB", "/nonexistent/libmanta_hip.so")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    with pytest.raises(RuntimeError, match="not found"):
        _lib.get()
    _lib.reset()


def test_package_never_references_the_oracle():
    """the product package must not import / load anything under oracle/"""
    root = os.path.join(util.ROOT, "mantaflow_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "libmanta_oracle" not in txt and "libmanta_ref" not in txt, f
                assert "oracle/" not in txt.replace("``oracle/``", "") or f == "_lib.py", f
