"""Cross-image batching front-end: many images' points -> one stream of large GPU batches.

The reference walks images one at a time (scripts/build_feature_bucket.py:749-786): load, whole-image
reflect pad on the CPU, per-patch PIL crops, ToTensor/Normalize, forward at batch 10, ``.tolist()``.
Here an image is uploaded once, its patches are cut by ``mmc_crop_patches`` (index arithmetic, no padded
copy) straight into a device buffer shared by consecutive images, and the backbone runs whenever
``batch_patches`` patches are resident -- so small point counts per image (10-25 in the reference's
data) still produce full 256-patch passes.  Output order and grouping are the reference's: one
``(n_points, 1280)`` float32 array per image, rows in ``rowcols`` order, or ``ImageFeatures`` objects with
``get_array((row, col))`` (mermaid_classifier/pyspacer/annotation.py:250).
"""

from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .backbone import Backbone, PATCH, _current_stream_ptr
from .spacer_shim import ImageFeatures, PointFeatures


def check_extract_inputs(image: np.ndarray, rowcols: Sequence[Tuple[int, int]], name: str = "image") -> None:
    """pyspacer ``check_extract_inputs`` as the reference calls it (annotation.py:240): every point must lie
    inside the image; raises ValueError naming the offending point."""
    h, w = image.shape[:2]
    for r, c in rowcols:
        if not (0 <= int(r) < h and 0 <= int(c) < w):
            raise ValueError(f"{name}: point ({r}, {c}) is outside the {h}x{w} image")
    if h <= PATCH or w <= PATCH:
        raise ValueError(f"{name}: image {h}x{w} must exceed the {PATCH}-pixel crop in both dimensions")


class BatchedExtractor:
    """Feeds a ``Backbone`` with cross-image batches.  ``batch_patches`` bounds the device patch buffer."""

    def __init__(self, backbone: Backbone, batch_patches: int = 1024):
        import torch
        self.bb = backbone
        self.cap = int(batch_patches)
        self.dev = torch.device("cuda", backbone.device_index)
        # two patch buffers: while the backbone works through one, the next images are cut / uploaded into the other on a
        # separate copy stream (a buffer is refilled only after the pass that read it has finished: per-buffer event)
        self._bufs = [torch.empty((self.cap, PATCH, PATCH, 3), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self._free = [torch.cuda.Event(), torch.cuda.Event()]
        self._done = [torch.cuda.Event(), torch.cuda.Event()]
        self._host = [torch.empty((self.cap, backbone.feature_dim), dtype=torch.float32).pin_memory() for _ in range(2)]
        # one device feature buffer per slot: a pass sees the same (patches, features) pair every time its slot comes round, so the
        # library's HIP-graph cache keeps hitting (a fresh output tensor per flush would mint a new combination per pass)
        self._fdev = [torch.empty((self.cap, backbone.feature_dim), dtype=torch.float32, device=self.dev) for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(device=self.dev)
        self._buf = self._bufs[0]

    def _crop_into(self, image: np.ndarray, rowcols: np.ndarray, offset: int) -> None:
        n = rowcols.shape[0]
        dst = self._buf[offset:offset + n]
        _lib.check(_lib.lib().mmc_crop_patches(image.ctypes.data, image.shape[0], image.shape[1], rowcols.ctypes.data, n,
                                               dst.data_ptr(), _lib.MMC_IN_HOST, self.bb.device_index,
                                               int(self._copy_stream.cuda_stream)))

    def extract_images(self, images: Iterable[np.ndarray], rowcols_per_image: Iterable[Sequence[Tuple[int, int]]]) -> List[np.ndarray]:
        """-> one (n_points, 1280) float32 array per image (empty arrays for images without points)."""
        import torch
        out: List[Optional[np.ndarray]] = []
        pending: List[Tuple[int, int, int]] = []   # (image index, offset in buffer, n)
        fill = 0
        cur = 0                                     # buffer being filled
        in_flight = None                            # (device features, pending list) of the pass launched last
        compute = torch.cuda.current_stream(self.dev)
        self._buf = self._bufs[cur]
        self._copy_stream.wait_stream(compute)      # whatever wrote these buffers before is done before the first cut lands

        def collect(job):
            slot, count, items = job
            self._done[slot].synchronize()                       # this pass's D2H only -- not whatever was enqueued after it
            feats = self._host[slot][:count].numpy().copy()
            for idx, off, n in items:
                part = feats[off:off + n]
                out[idx] = part if out[idx] is None else np.concatenate([out[idx], part])

        def flush():
            nonlocal fill, pending, cur, in_flight
            if fill == 0:
                return
            compute.wait_stream(self._copy_stream)              # the cuts of this buffer have landed
            feats_dev = self.bb.extract(self._bufs[cur][:fill], out=self._fdev[cur][:fill])  # asynchronous on the compute stream
            self._free[cur].record(compute)
            self._host[cur][:fill].copy_(feats_dev, non_blocking=True)   # pinned: the D2H is stream-ordered too
            self._done[cur].record(compute)
            job = (cur, fill, pending)
            if in_flight is not None:
                collect(in_flight)                               # the previous pass; this one keeps running meanwhile
            in_flight = job
            fill, pending = 0, []
            cur ^= 1
            self._buf = self._bufs[cur]
            self._copy_stream.wait_event(self._free[cur])        # refill only after the pass that read this buffer

        for idx, (image, rowcols) in enumerate(zip(images, rowcols_per_image)):
            im = np.asarray(image)
            if im.ndim == 2:
                im = np.stack([im] * 3, axis=-1)
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] < 3:
                raise ValueError(f"image {idx}: expected uint8 (H,W,3); got {im.dtype} {im.shape}")
            im = np.ascontiguousarray(im[..., :3])
            rc = np.ascontiguousarray(np.asarray(list(rowcols), dtype=np.int32).reshape(-1, 2))
            if len(rc):
                check_extract_inputs(im, rc, name=f"image {idx}")
            out.append(None if len(rc) else np.zeros((0, self.bb.feature_dim), np.float32))
            start = 0
            while start < len(rc):                      # an image with more points than the buffer spans flushes
                take = min(len(rc) - start, self.cap - fill)
                if take == 0:
                    flush()
                    continue
                self._crop_into(im, rc[start:start + take], fill)
                pending.append((idx, fill, take))
                fill += take
                start += take
        flush()
        if in_flight is not None:
            collect(in_flight)
        return [o if o is not None else np.zeros((0, self.bb.feature_dim), np.float32) for o in out]

    def extract_image_features(self, images, rowcols_per_image) -> List[ImageFeatures]:
        rowcols_per_image = [list(rc) for rc in rowcols_per_image]
        feats = self.extract_images(images, rowcols_per_image)
        res = []
        for f, rc in zip(feats, rowcols_per_image):
            pfs = [PointFeatures(int(r), int(c), row.tolist()) for (r, c), row in zip(rc, f)]
            res.append(ImageFeatures(pfs, True, self.bb.feature_dim, len(pfs)))
        return res
