"""CPU restatement of the pyspacer glue around the backbone (crop + call + stack).

TEST INFRASTRUCTURE ONLY (see oracle/efficientnet_b0_ref.py header).
PARITY UNPINNED: pyspacer==0.14.0 is absent offline; these follow its published
behaviour as used at the reference call sites:

* ``crop_patches``: pyspacer ``spacer/extract_features_utils.py`` [RECALL R7] --
  reflect-pad the whole image by ``crop_size`` on every side, then slice the
  ``crop_size`` square whose top-left is ``(row - crop_size//2, col - crop_size//2)``
  in unpadded coordinates.  Callers: ``FeatureExtractor.__call__`` via
  ``scripts/build_feature_bucket.py:775`` and ``mermaid_classifier/pyspacer/annotation.py:241``.
* ``extract``: ``FeatureExtractor.__call__`` -> ``crop_patches`` ->
  ``patches_to_features`` (scripts/build_feature_bucket.py:415-446).
* ``stack_reference_features``: scripts/extract_reference_features.py:50-59.
"""

from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import efficientnet_b0_ref as bb


def crop_patches(image: np.ndarray, rowcols: Sequence[Tuple[int, int]], crop_size: int = 224) -> np.ndarray:
    """image: (H,W,3) u8 (or (H,W) greyscale, promoted to RGB) -> (N,crop,crop,3) u8."""
    im = np.asarray(image)
    if im.ndim == 2:
        im = np.stack([im] * 3, axis=-1)
    pad = crop_size
    padded = np.pad(im, ((pad, pad), (pad, pad), (0, 0)), mode="reflect")
    half = crop_size // 2
    out = np.empty((len(rowcols), crop_size, crop_size, 3), dtype=np.uint8)
    for i, (row, col) in enumerate(rowcols):
        r0 = int(row) + pad - half
        c0 = int(col) + pad - half
        out[i] = padded[r0:r0 + crop_size, c0:c0 + crop_size, :3]
    return out


def crop_patches_indexed(image: np.ndarray, rowcols, crop_size: int = 224) -> np.ndarray:
    """Same result by index arithmetic only (no whole-image pad): numpy 'reflect'
    maps index i<0 to -i and i>=n to 2(n-1)-i (no edge repeat).  This is the
    formulation the GPU crop kernel uses; kept here so the two can be checked
    against each other on CPU."""
    im = np.asarray(image)
    if im.ndim == 2:
        im = np.stack([im] * 3, axis=-1)
    h, w = im.shape[:2]
    half = crop_size // 2
    out = np.empty((len(rowcols), crop_size, crop_size, 3), dtype=np.uint8)

    def refl(idx, n):
        idx = np.where(idx < 0, -idx, idx)
        return np.where(idx >= n, 2 * (n - 1) - idx, idx)

    for i, (row, col) in enumerate(rowcols):
        rr = refl(np.arange(row - half, row - half + crop_size), h)
        cc = refl(np.arange(col - half, col - half + crop_size), w)
        out[i] = im[np.ix_(rr, cc)][..., :3]
    return out


def extract(net: bb.EfficientNetB0Ref, image: np.ndarray, rowcols, batch_size: int = 10) -> np.ndarray:
    """(row,col) points of one image -> (N,1280) fp32 feature rows, in rowcols order."""
    return bb.patches_to_features(net, crop_patches(image, rowcols, bb.CROP_SIZE), batch_size)


def stack_reference_features(per_image_features: List[np.ndarray]) -> np.ndarray:
    """File order then point order, float32 (N, dim)."""
    rows = [np.asarray(v, dtype=np.float32) for feats in per_image_features for v in feats]
    x = np.asarray(rows, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError(f"expected a 2-D feature matrix; got shape {x.shape}")
    return x
