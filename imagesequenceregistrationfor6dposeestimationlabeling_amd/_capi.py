"""ctypes binding of libisr_hip.so (the C ABI declared in include/isr_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
torch is imported first so that the library binds to the same libamdhip64 instance as torch —
streams and device pointers are then shared.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch  # noqa: F401  (must be loaded before libisr_hip.so: shared HIP runtime)

import os

_PKG = Path(__file__).resolve().parent
# ISR_HIP_LIB: tooling only (A/B timing of two builds of the library in one GPU session)
LIB_PATH = Path(os.environ["ISR_HIP_LIB"]) if os.environ.get("ISR_HIP_LIB") else _PKG / "libisr_hip.so"

ISR_OK = 0
ABI_VERSION = 5
DTYPE_BF16 = 0
DTYPE_F32 = 1
DTYPE_BF16_LOG2 = 2
DTYPE_BF16_LOG2_SCREENED = 3
# ISR_TUNE_* knobs of include/isr_hip.h
TUNE = {"nn_path": 0, "nn_filter": 1, "icp_warm": 2, "nn_plan_rq": 3, "nn_plan_blocks": 4,
        "nn_tile_st": 5, "nn_tile_sq": 6, "nn_tile_tb": 7, "ep_wsum_valu": 8, "k1_f32_chain": 9, "k1_split": 10}


class IsrError(RuntimeError):
    pass


_lib = None

_vp = C.c_void_p
_i = C.c_int
_sz = C.c_size_t
_d = C.c_double
_f = C.c_float
_u64 = C.c_uint64
_i64 = C.c_int64

# name -> (restype, argtypes); kept in one table so tests can check it against the header.
SIGNATURES = {
    "isr_abi_version": (_i, []),
    "isr_last_error": (C.c_char_p, []),
    "isr_device_count": (_i, []),
    "isr_tuning_set": (_i, [_i, _i]),
    "isr_tuning_get": (_i, [_i]),
    "isr_corr_argmax_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "isr_corr_argmax": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_corr_argmax_digits": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "isr_corr_argmax_phase": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "isr_corr_argmax_recheck_count": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp]),
    "isr_corr_argmax_recheck_count_f32": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp]),
    "isr_corr_argmax_screen_redone": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp]),
    "isr_corr_quantize_fp6": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "isr_corr_argmax_clock_mhz": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp]),
    "isr_adds_bounds": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _d, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "isr_corr_topk_workspace_bytes": (_sz, [_i, _i]),
    "isr_corr_topk": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_corr_logsoftmax_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "isr_corr_logsoftmax": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp]),
    "isr_ep_corr_matrices": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "isr_mask_bbox": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "isr_crop_normalize": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "isr_select_top_workspace_bytes": (_sz, [_i]),
    "isr_select_top": (_i, [_vp, _i, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_select_top_dev": (_i, [_vp, _i, _vp, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_select_top_batch_workspace_bytes": (_sz, [_i, _i]),
    "isr_select_top_batch": (_i, [_vp, _i, _i, _vp, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_select_top_batch_digits": (_i, [_vp, _i, _i, _vp, _d, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_gather_corr_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "isr_pnp_ransac_batch_workspace_bytes": (_sz, [_i, _i, _i]),
    "isr_pnp_ransac_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _f, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_prep_queries_workspace_bytes": (_sz, [_i, _i, _i]),
    "isr_prep_queries": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_prep_queries_batch_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "isr_prep_queries_batch": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_gather_corr": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "isr_p3p_all_roots": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "isr_pnp_ransac_workspace_bytes": (_sz, [_i, _i]),
    "isr_p3p_hypotheses": (_i, [_vp, _vp, _vp, _i, _vp, _i, _u64, _vp, _vp, _vp, _vp]),
    "isr_ransac_score": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_pnp_refine": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "isr_pnp_ransac": (_i, [_vp, _vp, _vp, _i, _vp, _i, _u64, _f, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                            _sz, _vp]),
    "isr_nn_batched_workspace_bytes": (_sz, [_i, _i, _i]),
    "isr_nn_batched": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _sz, _vp]),
    "isr_ep_prepare": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_ep_pool_corr": (_i, [_vp, _i, _i, _vp, _vp]),
    "isr_ep_patch_corr_workspace_bytes": (_sz, [_i, _i, _i]),
    "isr_ep_patch_corr": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "isr_ep_patch_corr_cells": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "isr_ep_sample_workspace_bytes": (_sz, [_i, _i]),
    "isr_ep_sample": (_i, [_vp, _vp, _i, _i, _d, _i, _u64, _vp, _vp, _sz, _vp]),
    "isr_ep_sample_direct": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _d, _i, _u64, _vp, _vp, _sz, _vp]),
    "isr_ep_sample_weights": (_i, [_vp, _vp, _i, _i, _d, _vp, _vp, _sz, _vp]),
    "isr_ep_p3p": (_i, [_vp, _i, _i, _vp, _vp, _i, _u64, _vp, _vp, _vp]),
    "isr_ep_prune": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _d, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "isr_zbuf_score_workspace_bytes": (_sz, [_i, _i]),
    "isr_zbuf_score": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_estimate_pose_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "isr_estimate_pose": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _d, _vp, _i, _i, _i, _d, _d, _i, _i, _i, _u64,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_zbuf_score_direct": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_refine_objective": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_refine_objective_full": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "isr_add_metric": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp]),
    "isr_icp_workspace_bytes": (_sz, [_i, _i]),
    "isr_icp_point_to_point": (_i, [_vp, _i, _vp, _i, _d, _i, _d, _d, _vp, _vp, _vp, _sz, _vp]),
    "isr_rel_pose_table": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
}


def lib() -> C.CDLL:
    """Load libisr_hip.so (once).  Raises IsrError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise IsrError(
            f"{LIB_PATH} is missing: build it with "
            "`python -m imagesequenceregistrationfor6dposeestimationlabeling_amd.build` "
            "(there is no CPU fallback)")
    try:
        L = C.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover
        raise IsrError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise IsrError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if L.isr_abi_version() != ABI_VERSION:
        raise IsrError(f"ABI version {L.isr_abi_version()} != {ABI_VERSION}")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != ISR_OK:
        msg = lib().isr_last_error()
        raise IsrError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int | None:
    """Device (or host) address of a tensor, None for None."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise IsrError(
                "libisr_hip operates on device tensors only (got a CPU tensor); "
                "there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise IsrError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise IsrError("no device tensor given")
    return dev
