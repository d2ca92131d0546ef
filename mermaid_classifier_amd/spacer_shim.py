"""Minimal stand-ins for the few pyspacer types the extractor contract touches, used only
when the real ``spacer`` package is not importable (it is absent offline).
With pyspacer installed, ``extractor.build_extractor_class()`` subclasses the real
``spacer.extractors.EfficientNetExtractor`` instead and none of this is used.

Shapes follow pyspacer 0.14.0 as recalled [RECALL R8, R9] and as exercised by the
reference: ``DataLocation(storage_type, key, bucket_name)``
(scripts/build_feature_bucket.py:505-515), ``ImageFeatures.point_features[i].data`` and
``ImageFeatures.get_array((row, col))`` (mermaid_classifier/pyspacer/annotation.py:250,
scripts/extract_reference_features.py:52-54).
"""

from __future__ import annotations

import io
import json
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class DataLocation:
    storage_type: str
    key: str
    bucket_name: Optional[str] = None

    def __post_init__(self):
        if self.storage_type not in ("filesystem", "memory", "s3", "url"):
            raise ValueError(f"unknown storage_type {self.storage_type!r}")
        if self.storage_type == "s3" and not self.bucket_name:
            raise ValueError("s3 DataLocation needs a bucket_name")


_MEMORY_STORE: Dict[str, bytes] = {}


def store_bytes(loc: DataLocation, data: bytes) -> None:
    if loc.storage_type == "filesystem":
        with open(loc.key, "wb") as f:
            f.write(data)
    elif loc.storage_type == "memory":
        _MEMORY_STORE[loc.key] = bytes(data)
    else:
        raise NotImplementedError(f"storage_type {loc.storage_type!r} needs the real pyspacer storage layer")


def load_bytes(loc: DataLocation) -> bytes:
    if loc.storage_type == "filesystem":
        with open(loc.key, "rb") as f:
            return f.read()
    if loc.storage_type == "memory":
        return _MEMORY_STORE[loc.key]
    raise NotImplementedError(f"storage_type {loc.storage_type!r} needs the real pyspacer storage layer")


@dataclass
class PointFeatures:
    row: Optional[int]
    col: Optional[int]
    data: List[float]


@dataclass
class ImageFeatures:
    point_features: List[PointFeatures]
    valid_rowcol: bool
    feature_dim: int
    npoints: int
    _lookup: Dict[Tuple[int, int], int] = field(default_factory=dict, repr=False, compare=False)

    def __post_init__(self):
        if self.valid_rowcol:
            self._lookup = {(pf.row, pf.col): i for i, pf in enumerate(self.point_features)}

    def get_array(self, rowcol: Tuple[int, int]) -> np.ndarray:
        if not self.valid_rowcol:
            raise ValueError("Method requires valid rows and columns")
        return np.array(self.point_features[self._lookup[tuple(rowcol)]].data)

    # .featurevector on-disk format [RECALL R9]: JSON of serialize()
    def serialize(self) -> dict:
        return {
            "point_features": [{"row": pf.row, "col": pf.col, "data": list(pf.data)} for pf in self.point_features],
            "valid_rowcol": self.valid_rowcol, "feature_dim": self.feature_dim, "npoints": self.npoints,
        }

    @classmethod
    def deserialize(cls, data: dict) -> "ImageFeatures":
        pfs = [PointFeatures(d.get("row"), d.get("col"), list(d["data"])) for d in data["point_features"]]
        return cls(pfs, bool(data["valid_rowcol"]), int(data["feature_dim"]), int(data["npoints"]))

    def store(self, loc: DataLocation) -> None:
        store_bytes(loc, json.dumps(self.serialize()).encode())

    @classmethod
    def load(cls, loc: DataLocation) -> "ImageFeatures":
        return cls.deserialize(json.loads(load_bytes(loc).decode()))


@dataclass
class ExtractFeaturesReturnMsg:
    extractor_loaded_remotely: bool
    runtime: float


class FeatureExtractorBase:
    """The slice of pyspacer ``FeatureExtractor`` the reference relies on: constructor with
    ``data_locations``, ``load_datastream(key)``, ``__call__(im, rowcols)``."""

    DATA_LOCATION_KEYS: Sequence[str] = ()
    CROP_SIZE = 224

    def __init__(self, data_locations: Dict[str, DataLocation], data_hashes: Optional[Dict[str, str]] = None):
        for key in self.DATA_LOCATION_KEYS:
            if key not in data_locations:
                raise ValueError(f"data_locations is missing the required key {key!r}")
        self.data_locations = data_locations
        self.data_hashes = data_hashes or {}

    def load_datastream(self, key: str):
        loc = self.data_locations[key]
        return io.BytesIO(load_bytes(loc)), loc.storage_type in ("s3", "url")

    def crop(self, im, rowcols):
        raise NotImplementedError

    def patches_to_features(self, patch_list):
        raise NotImplementedError

    @property
    def feature_dim(self) -> int:
        raise NotImplementedError

    def __call__(self, im, rowcols):
        t0 = time.time()
        patch_list = self.crop(im, rowcols)
        feats, remote = self.patches_to_features(patch_list)
        pfs = [PointFeatures(int(r), int(c), list(f)) for (r, c), f in zip(rowcols, feats)]
        return (ImageFeatures(pfs, True, len(feats[0]) if len(feats) else self.feature_dim, len(feats)),
                ExtractFeaturesReturnMsg(extractor_loaded_remotely=remote, runtime=time.time() - t0))
