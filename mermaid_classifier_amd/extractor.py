"""MI355X drop-in for the reference's extractor seam.

Mirrors ``_DeviceCachingExtractor`` built by ``_build_device_caching_extractor_class()``
(reference scripts/build_feature_bucket.py:375-448): same constructor
(``data_locations={"weights": loc}, device=..., batch_size=...``), same
``patches_to_features(patch_list) -> (list[list[float]], loaded_remote)``, same cached-net
behaviour, same exception conventions -- but the forward pass is the HIP library, and
``transformation()`` (ToTensor+Normalize) is folded into its stem kernel.

The stock-class shape used by ``AnnotationRun`` (mermaid_classifier/pyspacer/annotation.py:236-241),
``EfficientNetExtractor(data_locations={"weights": loc})`` followed by
``extractor(image, rowcols) -> (ImageFeatures, msg)`` and ``features.get_array((row, col))``,
is accepted too: ``device`` and ``batch_size`` are optional.

When pyspacer is importable the class subclasses ``spacer.extractors.EfficientNetExtractor``
(so pyspacer's message layer, storage and ``ImageFeatures`` are the real ones); otherwise it
sits on ``spacer_shim.FeatureExtractorBase``.

``device="cpu"`` is NOT a backend of this package: it exists only because the reference's numeric
gate (``verify_device_numerics``, scripts/build_feature_bucket.py:475-480) builds its CPU side with
``cls(..., device="cpu")`` from the very class factory the integration replaces.  With pyspacer
installed such an instance runs pyspacer's own torch-CPU network exactly as the reference's
``_DeviceCachingExtractor`` does (parent ``load_weights`` -> ``eval`` -> ``transformation()`` ->
``net.extract_features``) -- the reference's arithmetic, none of ours; without pyspacer it raises.
"""

from __future__ import annotations

import logging
from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from . import spacer_shim
from .backbone import Backbone, FEATURE_DIM, PATCH, crop_patches_device

_DEFAULT_MAX_BATCH = 256
logger = logging.getLogger(__name__)


def resolve_device(name: str) -> str:
    """``--device`` resolution with the reference's semantics
    (scripts/build_feature_bucket.py:358-372), minus mps/cpu: this backend only has the
    HIP path, which PyTorch-ROCm reports as "cuda"."""
    import torch

    if name in ("auto", "cuda", "hip"):
        if not torch.cuda.is_available():
            raise RuntimeError(f"--device {name} requested but torch.cuda.is_available() is False.")
        return "cuda"
    if name.startswith("cuda:"):
        if not torch.cuda.is_available():
            raise RuntimeError(f"--device {name} requested but torch.cuda.is_available() is False.")
        return name
    if name == "cpu":
        raise RuntimeError("--device cpu is not served by mermaid_classifier_amd: the CPU path is stock pyspacer "
                           "(this package has no CPU backend; only the numerics gate builds a cpu-side extractor, "
                           "and that needs pyspacer installed).")
    raise RuntimeError(f"--device {name} is not served by mermaid_classifier_amd (HIP only; use 'cuda').")


def _patches_to_array(patch_list) -> np.ndarray:
    """list[PIL.Image | ndarray] of 224x224 RGB -> contiguous (N,224,224,3) uint8."""
    if isinstance(patch_list, np.ndarray):
        arr = patch_list
    else:
        arr = np.stack([np.asarray(p) for p in patch_list]) if len(patch_list) else np.zeros((0, PATCH, PATCH, 3), np.uint8)
    if arr.ndim == 3 and arr.shape[0] and arr.shape[1:] == (PATCH, PATCH):
        arr = np.stack([arr] * 3, axis=-1)  # greyscale patches -> RGB
    if arr.dtype != np.uint8:
        raise ValueError(f"patches must be 8-bit RGB; got dtype {arr.dtype}")
    if arr.ndim != 4 or arr.shape[1:3] != (PATCH, PATCH) or arr.shape[3] < 3:
        raise ValueError(f"patches must be (N,{PATCH},{PATCH},3); got {arr.shape}")
    return np.ascontiguousarray(arr[..., :3])


def _make_class(base, have_spacer: bool):
    class MI355EfficientNetExtractor(base):  # type: ignore[misc, valid-type]
        """EfficientNet-B0 extractor whose forward pass runs on an MI355X through
        libmermaid_mi355.so.  The backbone handle (packed weights + workspace on the GPU)
        is built once on first use and cached for the life of the instance."""

        DATA_LOCATION_KEYS = ["weights"]
        CROP_SIZE = PATCH
        BATCH_SIZE = 10  # pyspacer default [RECALL R10]; only bounds the workspace here

        def __init__(self, *, device: str = "cuda", batch_size: Optional[int] = None, **kwargs: Any):
            super().__init__(**kwargs)
            self._device = device
            self._batch_size = int(batch_size) if batch_size is not None else None
            if self._batch_size is not None and self._batch_size < 1:
                raise ValueError(f"batch_size must be >= 1; got {batch_size}")
            self._cached_net: Optional[Any] = None       # Backbone, or pyspacer's torch net for device="cpu"
            self._cached_loaded_remote = False
            if str(device) == "cpu" and not have_spacer:
                raise RuntimeError("device='cpu' needs pyspacer: the CPU side of the numerics gate is pyspacer's own "
                                   "torch network, and mermaid_classifier_amd has no CPU backend.")

        @property
        def feature_dim(self) -> int:
            return FEATURE_DIM

        @property
        def _stock_cpu(self) -> bool:
            return have_spacer and str(self._device) == "cpu"

        def _ensure_net(self) -> Tuple[Any, bool]:
            # scripts/build_feature_bucket.py:402-413: build once, report loaded_remote once
            if self._cached_net is not None:
                return self._cached_net, False
            weights_ds, loaded_remote = self.load_datastream("weights")
            if self._stock_cpu:
                # the reference's own CPU path (:406-410): pyspacer builds its torch model and loads the state dict
                net = base.load_weights(weights_ds)
                net = net.to("cpu")
                net.eval()
                self._cached_net = net
                self._cached_loaded_remote = loaded_remote
                return net, loaded_remote
            max_batch = max(self._batch_size or 0, _DEFAULT_MAX_BATCH) if self._batch_size is None else self._batch_size
            net = Backbone(weights_ds, device=resolve_device(self._device), max_batch=max_batch)
            self._cached_net = net
            self._cached_loaded_remote = loaded_remote
            return net, loaded_remote

        def patches_to_features(self, patch_list: Any) -> Any:
            # scripts/build_feature_bucket.py:415-446
            net, loaded_remote = self._ensure_net()
            if self._stock_cpu:
                return self._stock_cpu_features(net, patch_list), loaded_remote
            arr = _patches_to_array(patch_list)
            feats = net.extract(arr)            # (N,1280) float32 on the host
            return feats.tolist(), loaded_remote

        def _stock_cpu_features(self, net, patch_list) -> List[List[float]]:
            """device="cpu" with pyspacer installed: pyspacer's transformation() and torch network on the CPU, batched as
            the reference batches (:420-437).  Only the numerics gate comes here."""
            import torch
            from spacer.extractors.torch_extractors import transformation  # type: ignore

            transformer = transformation()
            bs = self._batch_size or self.BATCH_SIZE
            feats: List[List[float]] = []
            for b in range(0, len(patch_list), bs):
                batch_t = torch.stack([transformer(i) for i in patch_list[b:b + bs]])
                with torch.no_grad():
                    out = net.extract_features(batch_t)
                feats.extend(out.detach().to("cpu").tolist())
            return feats

        # crop on the GPU when used through the shim base; with real pyspacer the parent's
        # __call__ does its own CPU crop_patches and hands PIL patches to patches_to_features.
        def crop(self, im, rowcols):
            return crop_patches_device(np.asarray(im), rowcols, device=resolve_device(self._device)).cpu().numpy()

        def __call__(self, im, rowcols):
            if have_spacer:
                return super().__call__(im, rowcols)
            rowcols = [tuple(int(v) for v in rc) for rc in rowcols]
            image = np.asarray(im)
            h, w = image.shape[:2]
            for r, c in rowcols:
                if not (0 <= r < h and 0 <= c < w):
                    raise ValueError(f"rowcol ({r},{c}) outside the {h}x{w} image")
            return super().__call__(image, rowcols)

    return MI355EfficientNetExtractor


_cls_cache = {}


def build_extractor_class():
    """Lazy, like the reference's builder (scripts/build_feature_bucket.py:375-378): pyspacer
    is imported only here, after the caller has set its SPACER_* environment."""
    if "cls" in _cls_cache:
        return _cls_cache["cls"]
    try:
        from spacer.extractors import EfficientNetExtractor as _Base  # type: ignore
        cls = _make_class(_Base, True)
    except ImportError:
        cls = _make_class(spacer_shim.FeatureExtractorBase, False)
    _cls_cache["cls"] = cls
    return cls


def EfficientNetExtractor(*, data_locations, device: str = "cuda", batch_size: Optional[int] = None, **kwargs):
    """Stock-constructor shape (annotation.py:236-238)."""
    return build_extractor_class()(data_locations=data_locations, device=device, batch_size=batch_size, **kwargs)


def verify_device_numerics(extractor: Any, weights_loc: Any, batch_size: int, device: str, n_patches: int = 8,
                           threshold: float = 0.999, *, cpu_features_fn: Optional[Callable] = None) -> None:
    """The reference's only numeric gate for this path, with the reference's signature, logging and return value
    (scripts/build_feature_bucket.py:451-502): device features vs CPU features on ``n_patches`` seed-42 random patches,
    RuntimeError if the minimum cosine similarity is below ``threshold``; a no-op for ``device == "cpu"``.

    The CPU side is built the way the reference builds it -- ``build_extractor_class()(data_locations={"weights":
    weights_loc}, device="cpu", batch_size=batch_size)`` -- which is pyspacer's own torch-CPU network and therefore needs
    pyspacer.  ``cpu_features_fn(list_of_patches) -> (N, dim)`` (keyword-only, not in the reference) substitutes another
    CPU side, e.g. the oracle in the tests."""
    if device == "cpu":
        return
    rng = np.random.default_rng(seed=42)
    arrays = [rng.integers(0, 255, (PATCH, PATCH, 3), dtype=np.uint8) for _ in range(n_patches)]
    try:
        from PIL import Image
        patches: List[Any] = [Image.fromarray(a) for a in arrays]     # what pyspacer's transformation() takes
    except ImportError:
        patches = arrays
    if cpu_features_fn is None:
        cls = build_extractor_class()
        cpu_extractor = cls(data_locations={"weights": weights_loc}, device="cpu", batch_size=batch_size)
        cpu_feats, _ = cpu_extractor.patches_to_features(patches)
    else:
        cpu_feats = cpu_features_fn(arrays)
    device_feats, _ = extractor.patches_to_features(patches)
    a = np.asarray(device_feats)
    b = np.asarray(cpu_feats)
    sims = (a * b).sum(axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-12)
    logger.info("Device numerics check (%s vs cpu, %d random patches): min_cos=%.6f median=%.6f max_abs_diff=%.4g",
                device, n_patches, float(sims.min()), float(np.median(sims)), float(np.abs(a - b).max()))
    if sims.min() < threshold:
        raise RuntimeError(
            f"Device numerics check FAILED on {device}: min cosine similarity {sims.min():.6f} < {threshold}. "
            f"The features would not be safe to mix with previously CPU-extracted ones.")
