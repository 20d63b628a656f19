"""CPU: the C-ABI library loads, exports every symbol include/dgcn.h declares, and rejects bad
arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from distgcn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "dgcn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgcn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libdgcn.so lacks %s" % name
    assert sorted(_lib.SYMBOLS) == declared, "distgcn_amd/_lib.py binding list differs from the header"
    assert lib.dgcn_version() == 100


def test_argument_errors_without_gpu():
    lib = _lib.load()
    assert lib.dgcn_supports_batch(None, None, 0, None, None, None, None, None) == -1
    assert b"null argument" in lib.dgcn_last_error()
    assert lib.dgcn_lgs_batch(None, None, None, None, 0, None, None, None, None, None, None, None, None) == -1
    assert lib.dgcn_spmm_batch(None, None, 0, 0, None, 0, 0, None, 0, None, 0, None, 0, None) == -1
    assert lib.dgcn_transform_batch(None, 0, 0.0, 0, 0, None, 0, None, 0, None) == -1
    assert lib.dgcn_spmm_split(32) == 1
    ms = C.c_double()
    n = C.c_int64()
    assert lib.dgcn_timing_read(b"spmm", C.byref(ms), C.byref(n)) == 0 and n.value == 0
    with pytest.raises(_lib.DgcnError):
        _lib.check(-1, "x")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libdgcn.so"))
    with pytest.raises(_lib.DgcnError, match="no CPU fallback"):
        _lib.load()


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from distgcn_amd.engine import Engine
    with pytest.raises(_lib.DgcnError, match="no CPU fallback"):
        Engine("cuda")
    from distgcn_amd import heuristics
    import scipy.sparse as sp
    with pytest.raises(_lib.DgcnError):
        heuristics.local_greedy_search(sp.csr_matrix((3, 3)), [1.0, 2.0, 3.0])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "distgcn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_header_is_plain_c():
    """include/dgcn.h compiles as C99 on its own: extern "C", plain pointers and sizes, no C++ or torch types."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dgcn.h")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_python_option_bits_equal_the_header():
    """The DGCN_RESIDUAL_* option bits and DGCN_FAULT_* status bits the Python layer passes / decodes are the header's."""
    import re
    from distgcn_amd import engine
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dgcn.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(DGCN_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)\b", text)}
    assert defs["DGCN_RESIDUAL_SCORES_GIVEN"] == engine.Engine.SCORES_GIVEN
    assert defs["DGCN_RESIDUAL_COMPLETE_BY_PRIORITY"] == engine.Engine.COMPLETE_BY_PRIORITY
    assert defs["DGCN_RESIDUAL_FINISH_SMALL"] == engine.Engine.FINISH_SMALL
    bits = [defs[k] for k in ("DGCN_RESIDUAL_SCORES_GIVEN", "DGCN_RESIDUAL_COMPLETE_BY_PRIORITY", "DGCN_RESIDUAL_FINISH_SMALL")]
    assert len(set(bits)) == 3 and all(b & (b - 1) == 0 for b in bits)  # distinct single bits
