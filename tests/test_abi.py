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


def test_ctypes_prototypes_match_the_header_arity_and_kinds():
    """Every binding in dasac_hip/lib.py has as many arguments as its declaration in include/dasac_hip.h, pointers where the header
    has pointers and floating-point scalars where it has them (a drifted ctypes signature silently corrupts a call)."""
    import ctypes as C
    from dasac_hip import lib as L
    txt = open(os.path.join(ROOT, "include", "dasac_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct.*?}\s*\w+;", "", txt, flags=re.S)
    decls = dict(re.findall(r"\b(dasac_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt))
    assert set(decls) == set(L.PROTOTYPES)
    for name, args in decls.items():
        params = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        _, argtypes = L.PROTOTYPES[name]
        assert len(argtypes) == len(params), (name, len(argtypes), len(params))
        for p, t in zip(params, argtypes):
            is_ptr = "*" in p or p.startswith("dasac_stream_t")
            if is_ptr:
                assert t in (C.c_void_p, C.c_char_p) or isinstance(t, type(C.POINTER(C.c_int))), (name, p, t)
            elif p.startswith(("float ", "double ")):
                assert t is (C.c_float if p.startswith("float ") else C.c_double), (name, p, t)
            else:
                assert t in (C.c_int, C.c_int64, C.c_size_t, C.c_uint64), (name, p, t)
