"""libvinet_hip.so loads (no GPU needed) and exports every symbol include/vinet_hip.h declares."""
import ctypes
import os
import re

from vinet_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vinet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vinet_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 30
    if not os.path.exists(L.LIB_PATH):
        from vinet_amd import build
        build.build(verbose=False)
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libvinet_hip.so does not export %s" % n
    assert set(names) == set(L.SIGNATURES), set(names) ^ set(L.SIGNATURES)
    bound = L.load()
    assert bound.vinet_abi_version() == L.ABI_VERSION


def test_invalid_descriptor_is_rejected_without_a_gpu():
    lib = L.load()
    d = L.CConvDesc()
    d.dtype = 7
    assert lib.vinet_conv3d(ctypes.byref(d), None) < 0
    assert b"dtype" in lib.vinet_last_error()


def test_product_has_no_cpu_fallback():
    """a CPU tensor must raise, not silently compute somewhere else"""
    import pytest
    import torch
    from vinet_amd import loss, model_utils
    assert not L.is_test_double()
    with pytest.raises(Exception):
        model_utils.BasicConv3d(16, 32, 1, 1)(torch.zeros(1, 16, 1, 2, 2))
    with pytest.raises(Exception):
        loss.kldiv(torch.rand(1, 4, 4), torch.rand(1, 4, 4))
