"""ctypes binding of libdgcn.so (include/dgcn.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing the
import fails loudly with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGCN_LIB selects another build of the same library (e.g. the -DDGCN_DIAG profiling build)
LIB_PATH = os.environ.get("DGCN_LIB") or os.path.join(_HERE, "libdgcn.so")

OK = 0
ACT_LINEAR, ACT_LEAKY_RELU, ACT_RELU = 0, 1, 2
FAULT_SELF_LOOP, FAULT_NAN_PRIORITY, FAULT_DEGREE_RANGE, FAULT_BAD_COLUMN = 1, 2, 4, 8
FAULT_NAMES = {
    FAULT_SELF_LOOP: "adjacency has a self-loop (heuristics.py:94 would never terminate)",
    FAULT_NAN_PRIORITY: "NaN priority (heuristics.py:103-111 would never terminate)",
    FAULT_DEGREE_RANGE: "vertex degree outside the d^-1/2 table, or a row with more entries than its graph has vertices (repeated columns)",
    FAULT_BAD_COLUMN: "column index outside its graph's vertex range",
    16: "the workgroups of a graph lost each other (cluster variant of the fused kernel; now switched off for this process: call again)",
}

# every symbol include/dgcn.h declares; tests/test_cabi.py checks the library exports them all
SYMBOLS = (
    "dgcn_version", "dgcn_last_error", "dgcn_pack_measure", "dgcn_pack_batch", "dgcn_pack_compact_layout", "dgcn_pack_compact_batch", "dgcn_expand_compact_batch", "dgcn_supports_batch", "dgcn_supports2_count_batch", "dgcn_supports2_fill_batch", "dgcn_spmm_split", "dgcn_set_cluster", "dgcn_get_cluster", "dgcn_spmm_batch", "dgcn_spmm_f64acc_batch", "dgcn_transform_batch", "dgcn_transform_f64acc_batch",
    "dgcn_gcn_forward_workspace", "dgcn_gcn_forward_batch", "dgcn_gcn_forward_poly_batch", "dgcn_head_dual_batch", "dgcn_head_skip_batch", "dgcn_argmax_batch", "dgcn_lgs_batch", "dgcn_margin_risk_batch", "dgcn_lgs_masked_batch", "dgcn_solve_supported", "dgcn_solve_path", "dgcn_set_general", "dgcn_get_general", "dgcn_solve_workspace", "dgcn_solve_batch", "dgcn_solve_residual_batch",
    "dgcn_host_solver_create", "dgcn_host_solver_destroy", "dgcn_host_solver_submit", "dgcn_host_solver_result",
    "dgcn_timing_enable", "dgcn_timing_reset", "dgcn_timing_read", "dgcn_timing_sampling",
    "dgcn_set_option", "dgcn_get_option", "dgcn_option_count", "dgcn_option_name",
)


class DgcnError(RuntimeError):
    pass


class DgcnBatch(C.Structure):
    _fields_ = [
        ("num_graphs", C.c_int32), ("num_nodes", C.c_int32), ("num_edges", C.c_int32),
        ("max_nodes", C.c_int32), ("max_graph_edges", C.c_int32),
        ("graph_ptr", C.c_void_p), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p),
    ]


class DgcnCsr(C.Structure):
    _fields_ = [
        ("num_rows", C.c_int32), ("nnz", C.c_int32), ("max_graph_nnz", C.c_int32),
        ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("values", C.c_void_p),
    ]


class DgcnLayer(C.Structure):
    _fields_ = [
        ("in_dim", C.c_int32), ("out_dim", C.c_int32),
        ("weights", C.c_void_p), ("bias", C.c_void_p), ("act", C.c_int32),
    ]


class DgcnModel(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("num_supports", C.c_int32), ("layers_host", C.POINTER(DgcnLayer))]


class DgcnPackInfo(C.Structure):
    _fields_ = [("num_graphs", C.c_int32), ("num_nodes", C.c_int32), ("num_edges", C.c_int32), ("max_nodes", C.c_int32),
                ("max_graph_edges", C.c_int32), ("max_degree", C.c_int32), ("off_graph_ptr", C.c_int64),
                ("off_row_ptr", C.c_int64), ("off_col_idx", C.c_int64), ("off_weights", C.c_int64), ("total_bytes", C.c_int64)]


class DgcnCompactInfo(C.Structure):
    _fields_ = [("off_graph_ptr", C.c_int64), ("off_edge_ptr", C.c_int64), ("off_deg", C.c_int64), ("off_col", C.c_int64),
                ("off_weights", C.c_int64), ("total_bytes", C.c_int64)]


_lib = None


def load():
    """Return the loaded library (cached).  Raises DgcnError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DgcnError(
            "%s is missing: the HIP extension has not been built. Run "
            "`python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc, targets gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  It must be in the process first so
    # that libdgcn.so binds to THAT runtime; loading ours first would put two HIP runtimes in one
    # process and every launch would fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
    lib.dgcn_version.restype = C.c_int
    lib.dgcn_last_error.restype = C.c_char_p
    lib.dgcn_pack_measure.restype = C.c_int
    lib.dgcn_pack_measure.argtypes = [vp, vp, i32, i32, i32, C.POINTER(DgcnPackInfo), vp]
    lib.dgcn_pack_batch.restype = C.c_int
    lib.dgcn_pack_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp, sz, C.POINTER(DgcnPackInfo), i32]
    lib.dgcn_pack_compact_layout.restype = C.c_int
    lib.dgcn_pack_compact_layout.argtypes = [C.POINTER(DgcnPackInfo), C.POINTER(DgcnCompactInfo)]
    lib.dgcn_pack_compact_batch.restype = C.c_int
    lib.dgcn_pack_compact_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp, sz, C.POINTER(DgcnPackInfo), C.POINTER(DgcnCompactInfo), i32]
    lib.dgcn_expand_compact_batch.restype = C.c_int
    lib.dgcn_expand_compact_batch.argtypes = [vp, C.POINTER(DgcnCompactInfo), i32, i32, i32, vp, vp, vp]
    lib.dgcn_supports_batch.restype = C.c_int
    lib.dgcn_supports_batch.argtypes = [C.POINTER(DgcnBatch), vp, i32, vp, vp, vp, vp, vp]
    lib.dgcn_supports2_count_batch.restype = C.c_int
    lib.dgcn_supports2_count_batch.argtypes = [C.POINTER(DgcnBatch), vp, i32, vp, vp, vp]
    lib.dgcn_supports2_fill_batch.restype = C.c_int
    lib.dgcn_supports2_fill_batch.argtypes = [C.POINTER(DgcnBatch), vp, i32, vp, vp, vp, vp, vp]
    lib.dgcn_gcn_forward_poly_batch.restype = C.c_int
    lib.dgcn_gcn_forward_poly_batch.argtypes = [C.POINTER(DgcnBatch), C.POINTER(C.POINTER(DgcnCsr)), C.POINTER(DgcnModel),
                                                vp, f32, vp, vp, sz, vp]
    lib.dgcn_spmm_split.restype = C.c_int
    lib.dgcn_spmm_split.argtypes = [i32]
    lib.dgcn_spmm_batch.restype = C.c_int
    lib.dgcn_spmm_batch.argtypes = [C.POINTER(DgcnCsr), vp, i32, i32, vp, i32, i32, vp, i32, vp, i32, vp, i32, vp]
    lib.dgcn_transform_batch.restype = C.c_int
    lib.dgcn_transform_batch.argtypes = [vp, i32, f32, i32, i32, vp, i32, vp, i32, vp]
    lib.dgcn_set_cluster.restype = None
    lib.dgcn_set_cluster.argtypes = [i32]
    lib.dgcn_get_cluster.restype = i32
    lib.dgcn_get_cluster.argtypes = []
    lib.dgcn_transform_f64acc_batch.restype = C.c_int
    lib.dgcn_transform_f64acc_batch.argtypes = [vp, i32, f32, i32, i32, vp, i32, vp, i32, vp]
    lib.dgcn_spmm_f64acc_batch.restype = C.c_int
    lib.dgcn_spmm_f64acc_batch.argtypes = [C.POINTER(DgcnCsr), vp, i32, i32, vp, i32, i32, vp, i32, vp, i32, vp, i32, vp]
    lib.dgcn_gcn_forward_workspace.restype = sz
    lib.dgcn_gcn_forward_workspace.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnModel), i32]
    lib.dgcn_gcn_forward_batch.restype = C.c_int
    lib.dgcn_gcn_forward_batch.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnCsr), C.POINTER(DgcnModel),
                                           vp, f32, vp, vp, sz, i32, vp]
    lib.dgcn_head_dual_batch.restype = C.c_int
    lib.dgcn_head_dual_batch.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.dgcn_head_skip_batch.restype = C.c_int
    lib.dgcn_head_skip_batch.argtypes = [vp, f32, i32, vp, i32, vp, vp, i32, vp, vp]
    lib.dgcn_argmax_batch.restype = C.c_int
    lib.dgcn_argmax_batch.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.dgcn_lgs_batch.restype = C.c_int
    lib.dgcn_lgs_batch.argtypes = [C.POINTER(DgcnBatch), vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dgcn_margin_risk_batch.restype = C.c_int
    lib.dgcn_margin_risk_batch.argtypes = [C.POINTER(DgcnBatch), vp, vp, vp, vp, C.c_double, vp, vp]
    lib.dgcn_lgs_masked_batch.restype = C.c_int
    lib.dgcn_lgs_masked_batch.argtypes = [C.POINTER(DgcnBatch), vp, C.c_int64, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.dgcn_solve_supported.restype = C.c_int
    lib.dgcn_solve_supported.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnModel)]
    lib.dgcn_solve_path.restype = C.c_int
    lib.dgcn_solve_path.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnModel)]
    lib.dgcn_set_general.restype = None
    lib.dgcn_set_general.argtypes = [i32]
    lib.dgcn_get_general.restype = i32
    lib.dgcn_get_general.argtypes = []
    lib.dgcn_solve_batch.restype = C.c_int
    lib.dgcn_solve_workspace.restype = C.c_size_t
    lib.dgcn_solve_workspace.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnModel)]
    lib.dgcn_solve_batch.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnModel), vp, i32, vp, f32, vp, i32,
                                     vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.dgcn_solve_residual_batch.restype = C.c_int
    lib.dgcn_solve_residual_batch.argtypes = [C.POINTER(DgcnBatch), C.POINTER(DgcnModel), vp, i32, vp, f32, i32, vp, i32,
                                              i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.dgcn_host_solver_create.restype = C.c_int
    lib.dgcn_host_solver_create.argtypes = [C.POINTER(DgcnModel), vp, i32, i32, f32, i32, i32, i32, C.POINTER(vp)]
    lib.dgcn_host_solver_destroy.restype = None
    lib.dgcn_host_solver_destroy.argtypes = [vp]
    lib.dgcn_host_solver_submit.restype = C.c_int
    lib.dgcn_host_solver_submit.argtypes = [vp, vp, vp, vp, vp, i32, i32]
    lib.dgcn_host_solver_result.restype = C.c_int
    lib.dgcn_host_solver_result.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32),
                                            C.POINTER(i32), C.POINTER(i32)]
    lib.dgcn_timing_enable.restype = C.c_int
    lib.dgcn_timing_enable.argtypes = [i32]
    lib.dgcn_timing_reset.restype = C.c_int
    lib.dgcn_timing_read.restype = C.c_int
    lib.dgcn_timing_read.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.dgcn_timing_sampling.restype = i32
    lib.dgcn_timing_sampling.argtypes = []
    lib.dgcn_set_option.restype = C.c_int
    lib.dgcn_set_option.argtypes = [C.c_char_p, C.c_int64]
    lib.dgcn_get_option.restype = C.c_int
    lib.dgcn_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    lib.dgcn_option_count.restype = C.c_int
    lib.dgcn_option_count.argtypes = []
    lib.dgcn_option_name.restype = C.c_char_p
    lib.dgcn_option_name.argtypes = [i32, C.POINTER(C.c_int64)]
    _lib = lib
    # DGCN_OPTIONS="key=value,key=value": the library itself reads no environment variable (include/dgcn.h "Options");
    # this is the host layer's convenience for child processes of tests and A/B scripts.
    for kv in filter(None, (os.environ.get("DGCN_OPTIONS") or "").split(",")):
        key, _, val = kv.partition("=")
        set_option(key.strip(), int(val, 0))
    return lib


def set_option(key: str, value: int) -> None:
    """dgcn_set_option: one of the library's process-wide path switches (include/dgcn.h "Options")."""
    check(load().dgcn_set_option(key.encode(), int(value)), "dgcn_set_option(%s)" % key)


def get_option(key: str) -> int:
    out = C.c_int64(0)
    check(load().dgcn_get_option(key.encode(), C.byref(out)), "dgcn_get_option(%s)" % key)
    return int(out.value)


def option_defaults() -> dict:
    """{key: default} of every option the library knows."""
    lib, out = load(), {}
    for i in range(lib.dgcn_option_count()):
        d = C.c_int64(0)
        name = lib.dgcn_option_name(i, C.byref(d))
        out[name.decode()] = int(d.value)
    return out


class options:
    """Context manager: set options for a block, put the previous values back afterwards (tests, A/B scripts)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def check(rc: int, what: str = "libdgcn") -> None:
    if rc != OK:
        msg = load().dgcn_last_error()
        raise DgcnError("%s failed (%d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else "?"))


def fault_text(bits: int) -> str:
    return "; ".join(txt for bit, txt in FAULT_NAMES.items() if bits & bit) or "unknown fault %d" % bits
