"""ctypes binding of libngcf_hip.so (C ABI in include/ngcf_hip.h).

There is no fallback: if the library is missing or fails to load, importing the engine raises.
torch is imported first so that the library binds to the HIP runtime torch already loaded
(same SONAME libamdhip64.so.7) - device pointers and streams are then shared.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import torch  # noqa: F401  (must be loaded before libngcf_hip.so, see module docstring)

from . import _build

OK, ERR_ARG, ERR_HIP, ERR_INDEX, ERR_WORKSPACE = 0, 1, 2, 3, 4
ABI_VERSION = 7          # NGCF_ABI_VERSION of include/ngcf_hip.h these prototypes were written against

_vp, _i64, _i32, _f32, _u64 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64

# name -> (restype, argtypes); mirrors include/ngcf_hip.h one to one
PROTOTYPES = {
    "ngcf_last_error": (C.c_char_p, []),
    "ngcf_target_arch": (C.c_char_p, []),
    "ngcf_version": (C.c_int, []),
    "ngcf_options_from_env": (C.c_int, []),
    "ngcf_set_option": (C.c_int, [C.c_char_p, _i64]),
    "ngcf_set_option_str": (C.c_int, [C.c_char_p, C.c_char_p]),
    "ngcf_prof_enable": (C.c_int, [C.c_int]),
    "ngcf_prof_collect": (C.c_int, [C.POINTER(_i64), C.POINTER(C.c_double)]),
    "ngcf_csr_from_coo": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.POINTER(_vp), _vp]),
    "ngcf_csr_from_arrays": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.POINTER(_vp), _vp]),
    "ngcf_csr_plan": (C.c_int, [_vp, _i32, _vp]),
    "ngcf_csr_set_mode": (C.c_int, [_vp, C.c_int, _vp]),
    "ngcf_csr_filter": (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(_vp), _vp]),
    "ngcf_csr_filter_pos": (_vp, [_vp]),
    "ngcf_csr_filter_remap": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ngcf_csr_free": (None, [_vp]),
    "ngcf_csr_nnz": (_i64, [_vp]),
    "ngcf_csr_n_rows": (_i64, [_vp]),
    "ngcf_csr_n_cols": (_i64, [_vp]),
    "ngcf_csr_n_segments": (_i64, [_vp]),
    "ngcf_csr_max_row_len": (_i64, [_vp]),
    "ngcf_csr_swept_rows": (_i64, [_vp]),
    "ngcf_csr_rowptr": (_vp, [_vp]),
    "ngcf_csr_colidx": (_vp, [_vp]),
    "ngcf_csr_vals": (_vp, [_vp]),
    "ngcf_spmm_workspace_bytes": (_i64, [_vp, C.c_int]),
    "ngcf_layer_workspace_bytes": (_i64, [_vp, C.c_int, C.c_int]),
    "ngcf_dense_workspace_bytes": (_i64, [C.c_int, C.c_int]),
    "ngcf_spmm_product_width": (C.c_int, [_vp, _vp, _i64, C.c_int]),
    "ngcf_spmm_csr_f32": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _i64, _vp, _i64, _vp]),
    "ngcf_spmm_csr_dropout_f32": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _i64, _f32, C.POINTER(_u64), C.c_int, C.c_int,
                                            _vp, _i64, _vp]),
    "ngcf_layer_fused_f32": (C.c_int, [_vp, _vp, _i64, _vp, _i64, C.c_int, _vp, _vp, _vp, _vp, C.c_int,
                                       _f32, _f32, _u64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "ngcf_layer_dense_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp, C.c_int,
                                       _f32, _f32, _u64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "ngcf_copy_rows_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, C.c_int, _vp]),
    "ngcf_copy_rows2_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, C.c_int, _vp]),
    "ngcf_copy_rows_indexed_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, C.c_int, _vp]),
    "ngcf_feature_inject_f32": (C.c_int, [_vp, _i64, _i64, C.c_int, C.POINTER(_vp), C.POINTER(_vp),
                                          C.POINTER(_i64), C.c_int, _vp, _i64, C.c_double, _vp, _vp, _vp]),
    "ngcf_seeds_advance": (C.c_int, [_vp, C.c_int, _vp]),
    "ngcf_gather_rows_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "ngcf_gather_rows3_f32": (C.c_int, [_vp, _i64, C.c_int] + [_vp, _i64, _i64, _i64, _vp] * 3 + [_i64, _vp, _vp]),
    "ngcf_bpr_workspace_bytes": (_i64, [_i64]),
    "ngcf_bpr_fused_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, C.c_int, _f32, _f32, _vp, _vp, _i64, _vp]),
    "ngcf_bpr_backward_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, C.c_int, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ngcf_rows_sort_unique": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ngcf_segment_sum_rows_f32": (C.c_int, [_vp, _i64, C.c_int, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp]),
    "ngcf_layer_bwd_pre_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, C.c_int, _f32, _f32, _u64, _vp, _i64,
                                         _vp, _vp, _i64, _vp]),
    "ngcf_spmm_t_rows_f32": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp, _i64, _vp, _i64, _f32, C.POINTER(_u64), C.c_int, _vp, _i64, _vp]),
    "ngcf_layer_bwd_input_workspace_bytes": (_i64, [C.c_int]),
    "ngcf_layer_bwd_input_f32": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, _vp, C.c_int, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64,
                                           _vp, _i64, _vp]),
    "ngcf_bwd_weight_workspace_bytes": (_i64, []),
    "ngcf_layer_bwd_weight_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, C.c_int, C.c_int, _vp, _i64, _vp, _i64, _vp, _vp,
                                            _vp, _i64, _vp]),
    "ngcf_add_rows_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, C.c_int, _vp]),
    "ngcf_topk_rows_f32": (C.c_int, [_vp, _i64, _i64, _i64, C.c_int, _vp, _vp, _vp]),
    "ngcf_recommend_topk_f32": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, C.c_int, C.c_int, _vp, _i64, _vp, _vp, _vp]),
    "ngcf_shard_plan": (C.c_int, [C.POINTER(_i64), _i64, _i64, C.c_int, C.POINTER(_i64)]),
    "ngcf_allgather_rows": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp]),
    "ngcf_comm_size": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ngcf_p2p_create": (C.c_int, [C.c_int, C.c_int, _i64, C.c_char_p, C.POINTER(_vp)]),
    "ngcf_p2p_destroy": (None, [_vp]),
    "ngcf_p2p_handle": (C.c_int, [_vp, _vp]),
    "ngcf_p2p_connect": (C.c_int, [_vp, _vp]),
    "ngcf_p2p_local": (_vp, [_vp]),
    "ngcf_p2p_bytes": (_i64, [_vp]),
    "ngcf_p2p_publish": (C.c_int, [_vp, C.c_int, _u64, _vp]),
    "ngcf_p2p_pull": (C.c_int, [_vp, C.c_int, C.c_int, _u64, _i64, _vp, _i64, C.c_double]),
    "ngcf_p2p_ack": (C.c_int, [_vp, C.c_int, C.c_int, _u64]),
    "ngcf_p2p_wait_acks": (C.c_int, [_vp, C.c_int, _u64, C.c_double]),
    "ngcf_p2p_fence": (C.c_int, [_vp, _vp]),
    "ngcf_p2p_join": (C.c_int, [_vp, _vp]),
    "ngcf_torch_cpu_bernoulli": (C.c_int, [_vp, _i64, _i64, C.c_double, _vp, _vp, _f32, C.POINTER(_i64)]),
    "ngcf_torch_cpu_bernoulli_seq": (C.c_int, [_vp, _i64, C.c_int, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(_vp), C.POINTER(_vp),
                                               C.POINTER(_f32), C.POINTER(_i64)]),
    "ngcf_p2p_stats": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64), C.c_int]),
    "ngcf_sum_slots_f32": (C.c_int, [_vp, _i64, C.c_int, _i64, _vp, _vp]),
}

_lib = None


def lib_path() -> str:
    return _build.LIB


def load():
    """Load libngcf_hip.so.  The library is built by `python __graft_entry__.py build` (before any GPU or profiler
    start).  If it is missing or its sources changed since, it is rebuilt here under a file lock - unless NGCF_NO_BUILD=1
    (set it under rocprofv3 and wherever a compiler must not be spawned): then a missing or stale library raises
    (NGCF_ALLOW_STALE=1: a stale one is loaded with a loud warning, after the ABI version check).  A failed rebuild always raises; a stale library is never used silently."""
    global _lib
    if _lib is not None:
        return _lib
    if _build.needs_build():
        if os.environ.get("NGCF_NO_BUILD") == "1":
            if not os.path.exists(_build.LIB):
                raise RuntimeError(f"{_build.LIB} is not built and NGCF_NO_BUILD=1; run `python __graft_entry__.py build`. "
                                   "This package has no CPU or PyTorch fallback.")
            # a stale library may have another ABI (shifted arguments = out-of-bounds GPU accesses, not just old results)
            if os.environ.get("NGCF_ALLOW_STALE") != "1":
                raise RuntimeError(f"{_build.LIB} was built from other sources than the ones in the tree and NGCF_NO_BUILD=1 "
                                   "forbids rebuilding it here; run `python __graft_entry__.py build` first "
                                   "(NGCF_ALLOW_STALE=1 loads it anyway, after the ABI version check).")
            sys.stderr.write(f"[ngcf] WARNING: {_build.LIB} is OLDER than its sources (NGCF_NO_BUILD=1, NGCF_ALLOW_STALE=1): "
                             "results come from the stale library. Run `python __graft_entry__.py build`.\n")
        else:
            try:
                _build.build()
            except Exception as exc:  # noqa: BLE001
                raise RuntimeError(
                    "libngcf_hip.so is missing or older than its sources and could not be rebuilt here: " + str(exc)
                    + "\nThis package has no CPU or PyTorch fallback; run `python __graft_entry__.py build`.") from exc
    try:
        lib = C.CDLL(_build.LIB, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise RuntimeError(f"cannot load {_build.LIB}: {exc}; there is no fallback path") from exc
    lib.ngcf_version.restype = C.c_int
    if int(lib.ngcf_version()) != ABI_VERSION:
        raise RuntimeError(f"{_build.LIB} has ABI version {int(lib.ngcf_version())}, this package binds version {ABI_VERSION}: "
                           "rebuild it with `python __graft_entry__.py build`")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)        # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_option(name: str, value) -> None:
    """One tunable of the kernel dispatch by name (include/ngcf_hip.h, ngcf_set_option): tests and tools."""
    if isinstance(value, str):
        check(load().ngcf_set_option_str(name.encode(), value.encode()))
    else:
        check(load().ngcf_set_option(name.encode(), int(value)))


def options_from_env() -> None:
    """Re-read the NGCF_* variables (the library reads them once, on first use)."""
    check(load().ngcf_options_from_env())


def last_error() -> str:
    return load().ngcf_last_error().decode("utf-8", "replace")


def check(rc: int):
    """Translate a non-zero return code into the exception torch raises at the same call site."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INDEX:
        raise IndexError(msg)
    raise RuntimeError(msg)
