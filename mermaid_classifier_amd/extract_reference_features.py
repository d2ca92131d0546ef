"""Stack per-point vectors of ``.featurevector`` files into an ``(N, feature_dim)`` float32 ``.npy``.

Same behaviour and CLI as reference ``scripts/extract_reference_features.py:40-60`` (file order, then
``point_features`` order), for feature files written by this package's extractor (JSON ``ImageFeatures``,
spacer_shim.ImageFeatures.store) -- so the reference's live parity gate
(tests/pyspacer/test_portable_artifact.py:162-218) can be fed from MI355X-extracted features.

    python -m mermaid_classifier_amd.extract_reference_features --out reference_features.npy a.featurevector b.featurevector
"""

from __future__ import annotations

import argparse
from typing import Any, List, Sequence

import numpy as np

from .spacer_shim import DataLocation, ImageFeatures


def stack_feature_files(paths: Sequence[str]) -> np.ndarray:
    vectors: List[Any] = []
    for loc in paths:
        feats = ImageFeatures.load(DataLocation("filesystem", key=loc))
        for pf in feats.point_features:
            vectors.append(pf.data)
    x = np.asarray(vectors, dtype=np.float32)
    if x.ndim != 2:
        raise SystemExit(f"expected a 2-D feature matrix; got shape {x.shape}")
    return x


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--out", required=True, help="output .npy path")
    ap.add_argument("features", nargs="+", help="feature files (.featurevector), local paths")
    args = ap.parse_args(argv)
    x = stack_feature_files(args.features)
    np.save(args.out, x)
    print(f"wrote {x.shape[0]} feature vectors (dim {x.shape[1]}) to {args.out}")


if __name__ == "__main__":
    main()
