"""ctypes binding of libvlsa_hip.so -- the ONLY compute backend of this package.

There is no CPU / eager-PyTorch fallback: if the shared library is missing or a call fails, the
caller gets a ``VlsaNativeError`` immediately (the product path must fail loudly, never silently
route around the HIP kernels).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint32, c_void_p

_LIB_PATH = os.environ.get("VLSA_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib",
                                                             "libvlsa_hip.so")
ABI_VERSION = 1

# mirrors include/vlsa_hip.h
DT_F32, DT_BF16 = 0, 1
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_MFMA, KERNEL_DMA = 0, 1, 2, 3
POOL_MEAN, POOL_MAX, POOL_WEIGHT, POOL_GIVEN = 0, 1, 2, 3
MAX_P, MAX_K, MAX_D = 16, 64, 1024
TT_PERSISTENT = 0x100      # vlsa_tt_forward: or-ed into save_for_backward (VLSA_TT_PERSISTENT)
P_STRIDE = 16


ADAM_MAX_TENSORS = 16


class AdamTensor(ctypes.Structure):        # vlsa_adam_tensor
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_int64), ("hyper", c_int)]


class VlsaNativeError(RuntimeError):
    pass


_SIGNATURES = {
    "vlsa_abi_version": (c_int, []),
    "vlsa_error_string": (c_char_p, [c_int]),
    "vlsa_num_partials": (c_int, [c_int64]),
    "vlsa_qprep_bytes": (c_size_t, [c_int]),
    "vlsa_qprep_qeff": (c_void_p, [c_void_p, c_int]),
    "vlsa_qprep_qhat": (c_void_p, [c_void_p, c_int]),
    "vlsa_qprep_qnorm": (c_void_p, [c_void_p, c_int]),
    "vlsa_prepare_queries": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "vlsa_prepare_queries_and_text": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
                                              c_void_p, c_void_p, c_void_p]),
    "vlsa_vlfan_partial": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_vlfan_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "vlsa_vlfan_merge_strided": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_bwd_prep_bytes": (c_size_t, [c_int]),
    "vlsa_vlfan_backward": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_float, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_attn_normalise": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_normalize_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vlsa_head_workspace_bytes": (c_size_t, [c_int]),
    "vlsa_vlfan_merge_head": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 4 + [c_int]
                              + [c_void_p] * 12),
    "vlsa_head_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "vlsa_vlfan_forward_bag": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_int] + [c_void_p] * 13),
    "vlsa_fill_one_bag_tables": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "vlsa_vlfan_backward_bag": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_float] + [c_void_p] * 14 + [c_int]
                                + [c_void_p] * 10 + [c_int] + [c_void_p] * 3),
    "vlsa_batch_max_bags": (c_int, []),
    "vlsa_batch_forward_max_bags": (c_int, []),
    "vlsa_batch_partials_per_bag": (c_int, [c_int]),
    "vlsa_batch_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vlsa_vlfan_partial_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vlsa_batch_groups": (c_int, [c_void_p, c_int, c_int]),
    "vlsa_batch_partials_per_bag_ex": (c_int, [c_int, c_int, c_int]),
    "vlsa_vlfan_partial_batch_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "vlsa_vlfan_forward_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vlsa_vlfan_partial_batch_scores": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                                c_void_p]),
    "vlsa_attn_normalise_batch": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_vlfan_forward_batch_attn": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                              c_void_p, c_int64, c_void_p]),
    "vlsa_bwd_batch_partials": (c_int, []),
    "vlsa_vlfan_backward_bags": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vlsa_bwd_batch_prep_bytes": (c_size_t, [c_int, c_int]),
    "vlsa_vlfan_backward_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vlsa_vlfan_merge_batch_strided": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_vlfan_merge_head_batch_strided": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_head_forward_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "vlsa_pack_rows_bf16": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "vlsa_surv_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_float, c_int, c_int,
                               c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_surv_objective": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_float, c_float, c_int, c_int,
                                    c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "vlsa_query_chain": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vlsa_adam_step": (c_int, [c_void_p, c_int, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p]),
    "vlsa_gated_prep_bytes": (c_size_t, [c_int]),
    "vlsa_prepare_gated_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_void_p, c_void_p]),
    "vlsa_gated_scores": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vlsa_gated_scores_tiling": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "vlsa_gated_scores_big_tile": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "vlsa_gated_scores_pool_ws_floats": (c_int64, [c_int64]),
    "vlsa_gated_scores_pool": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_gated_scores_pool_adapter": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "vlsa_gated_scores_pool_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p]),
    "vlsa_gated_scores_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                        c_int64, c_void_p]),
    "vlsa_scored_pool_partial_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "vlsa_head_forward_batch_text": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_head_backward_batch": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_int] + [c_void_p] * 7),
    "vlsa_prompt_sentences": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_void_p, c_void_p]),
    "vlsa_prompt_sentences_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_void_p, c_void_p, c_void_p]),
    "vlsa_prototype_shapley": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "vlsa_featproj_prep_bytes": (c_size_t, []),
    "vlsa_prepare_featproj": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vlsa_feat_project": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "vlsa_adapter_head": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "vlsa_pool_num_partials": (c_int, [c_int64]),
    "vlsa_scored_pool_partial": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "vlsa_colmax": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vlsa_attn_scores": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "vlsa_query_pool_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_rowdot": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vlsa_scored_pool_backward": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "vlsa_topk_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "vlsa_topk_mean_ws": (c_int, [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "vlsa_topk_values": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vlsa_topk_mean_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vlsa_normalize_many": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "vlsa_topk_mean": (c_int, [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "vlsa_tt_workspace_bytes": (c_size_t, [c_void_p, c_void_p, c_int]),
    "vlsa_tt_status_offset": (c_int64, [c_void_p, c_void_p, c_int]),
    "vlsa_tt_packed_bytes": (c_size_t, [c_void_p, c_int]),
    "vlsa_tt_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "vlsa_tt_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "vlsa_tt_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "vlsa_tt_backward_train": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                       c_void_p]),
    "vlsa_mlp_bwd_tile_rows": (c_int, [c_int]),
    "vlsa_mlp_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vlsa_attn_scores_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_float, c_uint32, c_void_p]),
    "vlsa_attn_dx_prep_bytes": (c_size_t, [c_int]),
    "vlsa_prepare_attn_dx_weights": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "vlsa_attn_scores_backward_dx": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_float, c_uint32, c_void_p]),
    "vlsa_gated_scores_train": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_float, c_uint32, c_void_p]),
    "vlsa_feat_project_train": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_void_p]),
    "vlsa_feat_project_rowstats": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "vlsa_feat_project_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "vlsa_vlfan_backward_dx": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlsa_debug_probe": (c_int, [c_int, c_void_p, c_size_t, c_void_p]),
    "vlsa_xchg_max_peers": (c_int, []),
    "vlsa_xchg_result_floats": (c_size_t, [c_int, c_int, c_int, c_void_p]),
    "vlsa_xchg_alloc": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p]),
    "vlsa_xchg_open": (c_int, [c_void_p, c_void_p]),
    "vlsa_xchg_close": (c_int, [c_void_p]),
    "vlsa_xchg_free": (c_int, [c_void_p]),
    "vlsa_xchg_put": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_int64, c_void_p,
                              c_void_p]),
    "vlsa_xchg_wait": (c_int, [c_int, c_void_p, c_uint32, c_int64, c_void_p, c_void_p]),
    "vlsa_xchg_collect": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_uint32, c_int64]
                          + [c_void_p] * 9),
}

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def exported_symbols():
    """Names every build of the library must export (checked against include/vlsa_hip.h in the tests)."""
    return sorted(_SIGNATURES)


def load():
    """dlopen the library (once) and type every entry point.  Raises VlsaNativeError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise VlsaNativeError(
            f"{_LIB_PATH} is missing: build it with `python -m vlsa_amd.build` (hipcc, gfx950). "
            "vlsa_amd has no CPU fallback.")
    try:
        lib = ctypes.CDLL(_LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise VlsaNativeError(f"cannot load {_LIB_PATH}: {exc}") from exc
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise VlsaNativeError(f"{_LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    ver = lib.vlsa_abi_version()
    if ver != ABI_VERSION:
        raise VlsaNativeError(f"ABI mismatch: library {ver}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().vlsa_error_string(code).decode()
        raise VlsaNativeError(f"{what} failed: {msg} ({code})")


# ---- library-GEMM detours: an N-sized tensor (a bag's rows) that takes a torch route instead of a HIP kernel of this package ----
TORCH_ROUTE_ROWS = 1024          # P-sized inputs (<= 16 query rows) take torch ops by design; a bag is thousands of rows
torch_route_counts: dict = {}    # site -> number of calls with >= TORCH_ROUTE_ROWS rows (tests and benches read this)


def note_torch_route(site: str, rows: int, why: str) -> None:
    """Count -- and warn ONCE per site about -- a call that sends ``rows`` patch rows through torch library ops (rocBLAS GEMMs,
    [N, hidden] activations in HBM) because no fused kernel covers the configuration.  Correct, but several times slower than
    the HIP route and invisible otherwise."""
    if rows < TORCH_ROUTE_ROWS:
        return
    n = torch_route_counts.get(site, 0)
    torch_route_counts[site] = n + 1
    if n == 0:
        import warnings
        warnings.warn(f"vlsa_amd: {site}: {rows} patch rows go through torch library ops, not a fused HIP kernel ({why}); "
                      "further calls are counted in vlsa_amd._native.torch_route_counts", RuntimeWarning, stacklevel=3)


class TransientCaches:
    """nn.Module mixin.  The attributes named in ``_transient`` hold native handles (ctypes structures with device pointers), device
    scratch or cached results tied to them; none of it is state.  ``copy.deepcopy`` / ``pickle`` / ``torch.save(module)`` go
    through ``__getstate__``: the entries are dropped there (ctypes objects with pointers cannot even be pickled) and rebuilt on
    next use.  ``_transient`` maps a name to a zero-argument factory of its empty value, or to ``None`` = delete the attribute."""
    _transient: dict = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        for name, make in self._transient.items():
            if name in state:
                if make is None:
                    del state[name]
                else:
                    state[name] = make()
        return state
