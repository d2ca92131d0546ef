"""Backbone: thin handle wrapper over mmc_backbone_* (include/mmc.h).

torch is used only as a device-memory container (``tensor.data_ptr()``) and for the
current HIP stream; numpy arrays are passed as host pointers.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from . import _lib, weights as _weights

PATCH = 224
FEATURE_DIM = 1280


def _device_index(device) -> int:
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s in ("cuda", "hip"):
        try:
            import torch
            return torch.cuda.current_device()
        except Exception:
            return 0
    if ":" in s:
        return int(s.split(":")[1])
    raise ValueError(f"device {device!r} is not a HIP device (use 'cuda' or 'cuda:N')")


def _current_stream_ptr(device_index: int) -> int:
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.current_stream(device_index).cuda_stream)
    except Exception:
        pass
    return 0


class Backbone:
    """EfficientNet feature extractor resident on one MI355X.

    ``weights`` may be a path / byte stream of ``efficientnet.pt`` (pyspacer layout) or an
    already-loaded state dict ({key: array-like}, ``module.`` prefix optional).
    ``arch``: None = efficientnet-b0 for a checkpoint stream (what the reference path loads) and
    detected from the stem width for a state dict; "b0" / "b4" to insist."""

    def __init__(self, weights, device=0, max_batch: int = 256, arch=None, precision: str = "fp16"):
        """precision: "fp16" (fp16 storage and MFMA operands, fp32 accumulation) or "fp8" (EfficientNet-B4 only, BASELINE configs[4]: the
        7x7 stage's project convs on OCP e4m3 MFMA operands -- include/mmc.h MMC_PRECISION_FP8)."""
        if precision not in ("fp16", "fp8"):
            raise ValueError(f"precision must be 'fp16' or 'fp8', got {precision!r}")
        lib = _lib.lib()
        if isinstance(weights, dict):
            sd = {}
            for k, v in weights.items():
                k = k[7:] if k.startswith("module.") else k
                sd[k] = np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float64)
            A = _weights.get_arch(arch) if arch is not None else _weights.detect_arch(sd)
            want = _weights.expected_shapes(A)
            missing = sorted(k for k in want if k not in sd)
            if missing:
                raise _weights.WeightsError(f"state dict is missing keys: {missing[:10]}")
            bad = sorted(f"{k}: {tuple(sd[k].shape)} != {want[k]}" for k in want if tuple(sd[k].shape) != want[k])
            if bad:
                raise _weights.WeightsError(f"efficientnet-{A.name} state dict has wrong shapes: {bad[:10]}")
            blob = _weights.pack_backbone({k: sd[k] for k in want}, A)
        else:
            A = _weights.get_arch(arch)
            blob = _weights.pack_from_stream(weights, A)
        self.arch = A.name
        self.device_index = _device_index(device)
        self._h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self.precision = precision
        _lib.check(lib.mmc_backbone_create_ex(C.cast(buf, C.c_void_p), len(blob), A.arch_id, self.device_index,
                                              int(max_batch), _lib.MMC_PRECISION_FP8 if precision == "fp8" else 0, C.byref(self._h)))
        self.max_batch = int(max_batch)
        self.feature_dim = lib.mmc_feature_dim(self._h)
        self.lanes = lib.mmc_backbone_lanes(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().mmc_backbone_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(_lib.lib().mmc_backbone_workspace_bytes(self._h))

    def extract(self, patches, out=None):
        """patches: (N,224,224,3) uint8 -- numpy (host) or torch tensor on this device.
        Returns (N,1280) float32 of the same kind."""
        lib = _lib.lib()
        if isinstance(patches, np.ndarray):
            p = np.ascontiguousarray(patches)
            if p.dtype != np.uint8 or p.ndim != 4 or p.shape[1:] != (PATCH, PATCH, 3):
                raise ValueError(f"patches must be uint8 (N,{PATCH},{PATCH},3); got {p.dtype} {p.shape}")
            n = p.shape[0]
            res = np.empty((n, self.feature_dim), dtype=np.float32) if out is None else out
            if n:
                _lib.check(lib.mmc_backbone_extract(self._h, p.ctypes.data, n, res.ctypes.data,
                                                    _lib.MMC_IN_HOST | _lib.MMC_OUT_HOST,
                                                    _current_stream_ptr(self.device_index)))
            return res
        import torch
        if not isinstance(patches, torch.Tensor):
            raise TypeError("patches must be a numpy array or a torch tensor")
        if (not patches.is_cuda or patches.dtype != torch.uint8 or patches.dim() != 4
                or tuple(patches.shape[1:]) != (PATCH, PATCH, 3) or not patches.is_contiguous()):
            raise ValueError("device patches must be a contiguous uint8 cuda tensor (N,224,224,3)")
        if patches.device.index != self.device_index:
            raise ValueError(f"patches live on {patches.device}, backbone on device {self.device_index}")
        n = patches.shape[0]
        res = out if out is not None else torch.empty((n, self.feature_dim), dtype=torch.float32, device=patches.device)
        if n:
            _lib.check(lib.mmc_backbone_extract(self._h, patches.data_ptr(), n, res.data_ptr(), 0,
                                                _current_stream_ptr(self.device_index)))
        return res

    def read_activation(self, name: str, capacity: int) -> np.ndarray:
        out = np.empty(capacity, dtype=np.float32)
        nw = C.c_size_t(0)
        _lib.check(_lib.lib().mmc_backbone_read_activation(self._h, name.encode(), out.ctypes.data, capacity, C.byref(nw)))
        return out[:nw.value]

    def graph_stats(self) -> dict:
        """HIP-graph cache counters of this handle (include/mmc.h mmc_backbone_graph_stats)."""
        st = (C.c_int64 * 3)()
        _lib.check(_lib.lib().mmc_backbone_graph_stats(self._h, st))
        return {"captures": int(st[0]), "evictions": int(st[1]), "cached": int(st[2])}

    def profile(self, patches_dev, out_dev) -> List[Tuple[str, float]]:
        """One pass with HIP events around every launch -> [(launch name, ms)]."""
        cap = 128
        names = ((C.c_char * 64) * cap)()
        ms = (C.c_float * cap)()
        launches = (C.c_int * cap)()
        n_out = C.c_int(0)
        _lib.check(_lib.lib().mmc_backbone_profile(self._h, patches_dev.data_ptr(), patches_dev.shape[0],
                                                  out_dev.data_ptr(), _current_stream_ptr(self.device_index),
                                                  C.cast(names, C.c_void_p), ms, launches, cap, C.byref(n_out)))
        return [(names[i].value.decode(), float(ms[i])) for i in range(n_out.value)]


def crop_patches_device(image: np.ndarray, rowcols, device=0):
    """GPU crop_patches: (H,W,3) uint8 host image + [(row,col)] -> torch uint8 cuda tensor
    (N,224,224,3).  Replaces pyspacer ``crop_patches`` (reflect-pad 224 + slice)."""
    import torch
    im = np.asarray(image)
    if im.ndim == 2:
        im = np.stack([im] * 3, axis=-1)
    if im.ndim != 3 or im.shape[2] < 3 or im.dtype != np.uint8:
        raise ValueError(f"image must be uint8 (H,W,3); got {im.dtype} {im.shape}")
    im = np.ascontiguousarray(im[..., :3])
    rc = np.ascontiguousarray(np.asarray(rowcols, dtype=np.int32).reshape(-1, 2))
    n = rc.shape[0]
    di = _device_index(device)
    out = torch.empty((n, PATCH, PATCH, 3), dtype=torch.uint8, device=f"cuda:{di}")
    if n:
        _lib.check(_lib.lib().mmc_crop_patches(im.ctypes.data, im.shape[0], im.shape[1], rc.ctypes.data, n,
                                               out.data_ptr(), _lib.MMC_IN_HOST, di, _current_stream_ptr(di)))
    return out
