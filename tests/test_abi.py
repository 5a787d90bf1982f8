"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dasac_hip.h declares, and the ctypes table covers them all (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dasac_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dasac_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from dasac_hip import lib as L
    syms = declared_symbols()
    assert "dasac_pseudo_labels" in syms
    raw = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "missing export " + s
    assert sorted(L.PROTOTYPES) == syms
    assert L.load().dasac_version() == 1


def test_ops_refuse_cpu_tensors():
    import torch
    from dasac_hip import ops, DasacError
    with pytest.raises(DasacError):
        ops.pseudo_labels(torch.rand(1, 19, 4, 4), None, 0.75, 0.2)
