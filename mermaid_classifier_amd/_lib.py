"""ctypes binding of libmermaid_mi355.so (include/mmc.h).  There is no CPU fallback:
importing this module without the built library, or creating a handle without a HIP
device, raises."""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_NAME = "libmermaid_mi355.so"
LIB_PATH = _HERE / LIB_NAME

MMC_OK, MMC_ERR_ARG, MMC_ERR_WEIGHTS, MMC_ERR_HIP, MMC_ERR_NOMEM = 0, 1, 2, 3, 4
MMC_PRECISION_FP8 = 1   # include/mmc.h: flags of mmc_backbone_create_ex
MMC_IN_HOST, MMC_OUT_HOST = 1, 2

# every symbol include/mmc.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "mmc_last_error", "mmc_version", "mmc_device_count",
    "mmc_backbone_create", "mmc_backbone_create_ex", "mmc_fp8_e4m3_encode", "mmc_backbone_destroy", "mmc_feature_dim", "mmc_backbone_max_batch", "mmc_backbone_lanes",
    "mmc_backbone_workspace_bytes", "mmc_backbone_extract", "mmc_backbone_read_activation",
    "mmc_backbone_profile", "mmc_backbone_graph_stats", "mmc_crop_patches",
    "mmc_head_create", "mmc_head_destroy", "mmc_head_input_dim", "mmc_head_num_classes", "mmc_head_predict",
    "mmc_trainer_create", "mmc_trainer_destroy", "mmc_trainer_partial_fit", "mmc_trainer_partial_fit_ordered", "mmc_trainer_get_params", "mmc_trainer_adam_state",
    "mmc_trainer_logits",
    "mmc_dist_unique_id", "mmc_dist_create", "mmc_dist_destroy", "mmc_gather_features",
]


class LibraryMissingError(ImportError):
    pass


def _load() -> C.CDLL:
    path = Path(os.environ.get("MMC_LIBRARY", LIB_PATH))
    if not path.is_file():
        raise LibraryMissingError(
            f"{path} not found: build it with `python -m mermaid_classifier_amd.build` "
            "(hipcc --offload-arch=gfx950). mermaid_classifier_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not share
    # devices, so make sure torch's copy is the one already resident before our library resolves its
    # libamdhip64 dependency (otherwise /opt/rocm's would load first and torch tensors would live in a
    # different runtime than our kernels).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(path))
    vp, i32, i64, u32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint, C.c_size_t
    fp = C.POINTER(C.c_float)
    lib.mmc_last_error.restype = C.c_char_p
    lib.mmc_last_error.argtypes = []
    lib.mmc_version.restype = i32
    lib.mmc_device_count.restype = i32
    lib.mmc_backbone_create.restype = i32
    lib.mmc_backbone_create.argtypes = [vp, sz, i32, i32, i32, C.POINTER(vp)]
    lib.mmc_backbone_create_ex.restype = i32
    lib.mmc_backbone_create_ex.argtypes = [vp, sz, i32, i32, i32, u32, C.POINTER(vp)]
    lib.mmc_fp8_e4m3_encode.restype = i32
    lib.mmc_fp8_e4m3_encode.argtypes = [vp, vp, sz]
    lib.mmc_backbone_destroy.restype = None
    lib.mmc_backbone_destroy.argtypes = [vp]
    lib.mmc_feature_dim.restype = i32
    lib.mmc_feature_dim.argtypes = [vp]
    lib.mmc_backbone_max_batch.restype = i32
    lib.mmc_backbone_max_batch.argtypes = [vp]
    lib.mmc_backbone_lanes.restype = i32
    lib.mmc_backbone_lanes.argtypes = [vp]
    lib.mmc_backbone_workspace_bytes.restype = sz
    lib.mmc_backbone_workspace_bytes.argtypes = [vp]
    lib.mmc_backbone_extract.restype = i32
    lib.mmc_backbone_extract.argtypes = [vp, vp, i64, vp, u32, vp]
    lib.mmc_backbone_read_activation.restype = i32
    lib.mmc_backbone_read_activation.argtypes = [vp, C.c_char_p, vp, sz, C.POINTER(sz)]
    lib.mmc_backbone_profile.restype = i32
    lib.mmc_backbone_profile.argtypes = [vp, vp, i64, vp, vp, vp, fp, C.POINTER(i32), i32, C.POINTER(i32)]
    lib.mmc_backbone_graph_stats.restype = i32
    lib.mmc_backbone_graph_stats.argtypes = [vp, C.POINTER(i64)]
    lib.mmc_crop_patches.restype = i32
    lib.mmc_crop_patches.argtypes = [vp, i32, i32, vp, i64, vp, u32, i32, vp]
    lib.mmc_head_create.restype = i32
    lib.mmc_head_create.argtypes = [C.POINTER(fp), C.POINTER(fp), C.POINTER(i32), i32, fp, fp, i32, i32, C.POINTER(vp)]
    lib.mmc_head_destroy.restype = None
    lib.mmc_head_destroy.argtypes = [vp]
    lib.mmc_head_input_dim.restype = i32
    lib.mmc_head_input_dim.argtypes = [vp]
    lib.mmc_head_num_classes.restype = i32
    lib.mmc_head_num_classes.argtypes = [vp]
    lib.mmc_head_predict.restype = i32
    lib.mmc_head_predict.argtypes = [vp, vp, i64, vp, vp, u32, vp]
    f32 = C.c_float
    f64 = C.c_double
    lib.mmc_trainer_create.restype = i32
    lib.mmc_trainer_create.argtypes = [C.POINTER(fp), C.POINTER(fp), C.POINTER(i32), i32, f64, f64, f64, f64, f64, fp, i32, C.POINTER(vp)]
    lib.mmc_trainer_destroy.restype = None
    lib.mmc_trainer_destroy.argtypes = [vp]
    lib.mmc_trainer_partial_fit.restype = i32
    lib.mmc_trainer_partial_fit.argtypes = [vp, vp, vp, i64, i32, C.POINTER(C.c_double), vp]
    lib.mmc_trainer_partial_fit_ordered.restype = i32
    lib.mmc_trainer_partial_fit_ordered.argtypes = [vp, vp, vp, vp, i64, i32, C.POINTER(C.c_double), vp]
    lib.mmc_trainer_get_params.restype = i32
    lib.mmc_trainer_get_params.argtypes = [vp, C.POINTER(fp), C.POINTER(fp)]
    lib.mmc_trainer_adam_state.restype = i32
    lib.mmc_trainer_adam_state.argtypes = [vp, i32, i32, C.POINTER(fp), C.POINTER(fp), C.POINTER(C.c_longlong)]
    lib.mmc_trainer_logits.restype = i32
    lib.mmc_trainer_logits.argtypes = [vp, vp, i64, vp, vp]
    lib.mmc_dist_unique_id.restype = i32
    lib.mmc_dist_unique_id.argtypes = [vp]
    lib.mmc_dist_create.restype = i32
    lib.mmc_dist_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.mmc_dist_destroy.restype = None
    lib.mmc_dist_destroy.argtypes = [vp]
    lib.mmc_gather_features.restype = i32
    lib.mmc_gather_features.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def check(status: int) -> None:
    """Translate a status code into the Python exception the reference's callers expect
    (ValueError for bad shapes/URIs, RuntimeError for device problems)."""
    if status == MMC_OK:
        return
    msg = lib().mmc_last_error().decode("utf-8", "replace")
    if status in (MMC_ERR_ARG, MMC_ERR_WEIGHTS):
        raise ValueError(msg)
    if status == MMC_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)
