"""CPU-side checks of the drop-in boundary: the C-ABI library is present, loads, and exports
exactly the entry points include/badger_pf.h declares (no compute call is made here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "badger_pf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bpf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from badger_amcl_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_python_binding_covers_the_header():
    from badger_amcl_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_no_gpu_fails_loudly_not_silently():
    """Without a GPU bpf_create must return an error; with one it must succeed.  Either way the
    package never computes on the CPU."""
    import badger_amcl_amd as bpf
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        e = bpf.Engine(0)
        e.close()
    else:
        with pytest.raises(bpf.BpfError):
            bpf.Engine(0)


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "badger_amcl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inl", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "amcl_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
