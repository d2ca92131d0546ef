"""Per-source extraction driver: the caller of the hot path, with the reference's bookkeeping.

Mirrors ``process_source`` of the reference (scripts/build_feature_bucket.py:691-788) -- image ids in sorted
order, ``skipped`` for images without points or with an existing ``.featurevector`` (resume), ``ok`` /
``failed`` per image with the failure isolated to that image, a progress JSONL record and an error CSV row
per event (``record_progress`` / ``record_failure``, :794-823), ``RunCounters`` (:549-559) -- but instead of
one synchronous ``extract_features(msg)`` per image it

  * loads/decodes images on a small thread pool, ``prefetch`` images ahead of the GPU;
  * validates each image on the host first (``check_extract_inputs``), so a bad image is recorded as failed
    without ever entering a batch;
  * hands groups of images to ``BatchedExtractor.extract_image_features`` (crop on the GPU, cross-image
    256-patch passes) and writes every image's ``ImageFeatures`` through ``store_features``.

Storage stays behind two callables (``load_image(image_id) -> uint8 HxWx3`` and
``store_features(image_id, ImageFeatures)``): S3, listing and the annotations CSV are the reference's
control plane and are out of scope.  ``fs_store`` / ``fs_existing`` give the same key layout
(``s{source}/features/i{image}.featurevector``, build_feature_bucket.py:541) on a local directory.
"""

from __future__ import annotations

import csv
import json
import os
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from datetime import datetime, timezone
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Set, Tuple

import numpy as np

from .pipeline import check_extract_inputs
from .spacer_shim import DataLocation, ImageFeatures

RowCols = Sequence[Tuple[int, int]]


@dataclass
class RunCounters:
    """Same fields as the reference's RunCounters (build_feature_bucket.py:549-559) minus the annotation copies."""
    sources_done: int = 0
    sources_skipped: int = 0
    images_ok: int = 0
    images_skipped: int = 0
    images_failed: int = 0
    started: float = field(default_factory=time.monotonic)


def _ts() -> str:
    return datetime.now(timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ")


def record_progress(writer, source_id: str, image_id: str, outcome: str, **extra: object) -> None:
    """One JSON line per image event; fields as build_feature_bucket.py:794-807."""
    if writer is None:
        return
    writer.write(json.dumps({"ts": _ts(), "source_id": source_id, "image_id": image_id, "outcome": outcome, **extra}) + "\n")
    writer.flush()


def record_failure(writer, source_id: str, image_id: str, error_type: str, error_msg: str) -> None:
    """One CSV row per failure: ts, source, image, exception type, message (build_feature_bucket.py:810-823)."""
    if writer is None:
        return
    writer.writerow([_ts(), source_id, image_id, error_type, error_msg])


def feature_key(source_id: str, image_id: str) -> str:
    return f"s{source_id}/features/i{image_id}.featurevector"


def fs_store(root: str, source_id: str) -> Callable[[str, ImageFeatures], None]:
    def store(image_id: str, feats: ImageFeatures) -> None:
        path = os.path.join(root, feature_key(source_id, image_id))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        feats.store(DataLocation("filesystem", path))
    return store


def fs_existing(root: str, source_id: str) -> Set[str]:
    """Image ids that already have a feature file (the reference's ``list_existing_feature_image_ids``)."""
    d = os.path.join(root, f"s{source_id}", "features")
    if not os.path.isdir(d):
        return set()
    return {f[1:-len(".featurevector")] for f in os.listdir(d) if f.startswith("i") and f.endswith(".featurevector")}


def _as_image(arr: Any) -> np.ndarray:
    im = np.asarray(arr)
    if im.ndim == 2:
        im = np.stack([im] * 3, axis=-1)
    if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] < 3:
        raise ValueError(f"expected a uint8 (H,W,3) image; got {im.dtype} {im.shape}")
    return np.ascontiguousarray(im[..., :3])


def process_source(*, source_id: str, images: Mapping[str, RowCols], load_image: Callable[[str], Any],
                   store_features: Callable[[str, ImageFeatures], None], extractor: Any,
                   counters: Optional[RunCounters] = None, existing: Iterable[str] = (), dry_run: bool = False,
                   progress_writer=None, error_writer=None, prefetch: int = 4, group_images: int = 32) -> RunCounters:
    """Extract and store features for every image of one source.

    ``extractor`` is a ``pipeline.BatchedExtractor`` (anything with ``extract_image_features(images, rowcols)``).
    Outcomes, counters and log records follow the reference loop; only the execution order differs
    (grouped).  ``KeyboardInterrupt`` propagates, every other per-image exception is recorded and skipped."""
    counters = counters or RunCounters()
    existing = set(existing)
    if not images:
        counters.sources_skipped += 1
        return counters

    todo: List[str] = []
    for image_id in sorted(images.keys()):
        rowcols = images[image_id]
        if not rowcols:
            counters.images_skipped += 1
            record_progress(progress_writer, source_id, image_id, "skipped", reason="no_rowcols")
        elif image_id in existing:
            counters.images_skipped += 1
            record_progress(progress_writer, source_id, image_id, "skipped", reason="exists")
        elif dry_run:
            counters.images_ok += 1
            record_progress(progress_writer, source_id, image_id, "ok", dry_run=True)
        else:
            todo.append(image_id)

    def fail(image_id: str, exc: BaseException) -> None:
        counters.images_failed += 1
        record_failure(error_writer, source_id, image_id, type(exc).__name__, str(exc))
        record_progress(progress_writer, source_id, image_id, "failed", error_type=type(exc).__name__)

    def run_group(group: List[Tuple[str, np.ndarray, List[Tuple[int, int]]]]) -> None:
        if not group:
            return
        try:
            feats = extractor.extract_image_features([g[1] for g in group], [g[2] for g in group])
        except KeyboardInterrupt:
            raise
        except Exception as exc:       # a device-side failure takes the whole group with it
            for image_id, _, _ in group:
                fail(image_id, exc)
            return
        for (image_id, _, _), f in zip(group, feats):
            try:
                store_features(image_id, f)
                counters.images_ok += 1
                record_progress(progress_writer, source_id, image_id, "ok")
            except KeyboardInterrupt:
                raise
            except Exception as exc:
                fail(image_id, exc)

    group: List[Tuple[str, np.ndarray, List[Tuple[int, int]]]] = []
    with ThreadPoolExecutor(max_workers=max(1, prefetch)) as pool:
        window: List[Tuple[str, Any]] = []
        it = iter(todo)

        def refill() -> None:
            while len(window) < max(1, prefetch):
                nxt = next(it, None)
                if nxt is None:
                    return
                window.append((nxt, pool.submit(load_image, nxt)))

        refill()
        while window:
            image_id, fut = window.pop(0)
            refill()
            try:
                im = _as_image(fut.result())
                rc = [(int(r), int(c)) for r, c in images[image_id]]
                check_extract_inputs(im, rc, name=f"s{source_id} i{image_id}")
            except KeyboardInterrupt:
                raise
            except Exception as exc:
                fail(image_id, exc)
                continue
            group.append((image_id, im, rc))
            if len(group) >= group_images:
                run_group(group)
                group = []
        run_group(group)
    counters.sources_done += 1
    return counters


def open_logs(progress_path: Optional[str], error_path: Optional[str]):
    """-> (progress file or None, csv writer or None, closer).  The error CSV gets the reference's header."""
    pf = open(progress_path, "a") if progress_path else None
    ef = open(error_path, "a", newline="") if error_path else None
    ew = None
    if ef is not None:
        ew = csv.writer(ef)
        if ef.tell() == 0:
            ew.writerow(["ts", "source_id", "image_id", "error_type", "error_msg"])

    def close() -> None:
        for f in (pf, ef):
            if f is not None:
                f.close()
    return pf, ew, close
