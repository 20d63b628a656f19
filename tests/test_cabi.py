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


def test_options_are_the_only_process_state_and_no_environment_is_read():
    """dgcn_set_option / dgcn_get_option (include/dgcn.h "Options"): every key of csrc/options.h is listed by the library with
    its default, a set is seen by the next get, unknown keys are refused, the older setters are names of the same words - and
    no source of the library calls getenv (a stateless, re-entrant library; round-5 review item 6)."""
    lib = _lib.load()
    table = open(os.path.join(ROOT, "distgcn_amd", "csrc", "options.h")).read()
    declared = {m.group(1): int(m.group(2)) for m in re.finditer(r'X\(OPT_[A-Z0-9_]+,\s*"([a-z0-9_]+)",\s*(-?\d+)\)', table)}
    defaults = _lib.option_defaults()
    assert defaults == declared and len(defaults) == lib.dgcn_option_count() >= 30
    assert lib.dgcn_option_name(lib.dgcn_option_count(), None) is None and lib.dgcn_option_name(-1, None) is None
    before = {k: _lib.get_option(k) for k in defaults}
    try:
        for i, k in enumerate(defaults):
            _lib.set_option(k, 1000 + i)
        assert [_lib.get_option(k) for k in defaults] == [1000 + i for i in range(len(defaults))]
        _lib.set_option("diag_stamps", 0x7fff12345678)  # 64-bit values (a device address in the -DDGCN_DIAG builds)
        assert _lib.get_option("diag_stamps") == 0x7fff12345678
        lib.dgcn_set_cluster(4)
        assert _lib.get_option("fused_cluster") == 4 and lib.dgcn_get_cluster() == 4
        lib.dgcn_set_general(1)
        assert _lib.get_option("general") == 1 and lib.dgcn_get_general() == 1
        with _lib.options(wide1=0, tail=0):
            assert _lib.get_option("wide1") == 0 and _lib.get_option("tail") == 0
        assert _lib.get_option("wide1") == 1000 + list(defaults).index("wide1")  # (put back by the context manager)
    finally:
        for k, v in before.items():
            _lib.set_option(k, v)
    assert lib.dgcn_set_option(b"no_such_switch", 1) == -1 and b"unknown option" in lib.dgcn_last_error()
    assert lib.dgcn_get_option(b"fused_block", None) == -1
    assert lib.dgcn_set_option(None, 1) == -1
    csrc = os.path.join(ROOT, "distgcn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".c")):
            code = re.sub(r"//.*?$|/\*.*?\*/", "", open(os.path.join(csrc, f)).read(), flags=re.S | re.M)
            assert "getenv" not in code, f


def test_options_from_many_threads():
    """Set / get from eight threads at once: every read returns a value some thread wrote (atomic words, no torn state)."""
    import threading
    _lib.load()
    seen, stop = [], []

    def worker(i):
        for r in range(2000):
            _lib.set_option("lgs_lpv", i)
            v = _lib.get_option("lgs_lpv")
            if not 0 <= v < 8:
                seen.append(v)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    _lib.set_option("lgs_lpv", -1)
    assert not seen and not stop


def test_dgcn_options_environment_is_applied_by_the_python_layer():
    """DGCN_OPTIONS="key=value,..." is read by distgcn_amd/_lib.py at load (not by the library): a child process sees it."""
    import subprocess
    import sys
    code = "from distgcn_amd import _lib; _lib.load(); print(_lib.get_option('wide1'), _lib.get_option('fused_cluster'), _lib.get_option('tail'))"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DGCN_OPTIONS="wide1=0, fused_cluster=4"), cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split() == ["0", "4", "-1"], r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DGCN_OPTIONS="nonsense=1"), cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "unknown option" in r.stderr
