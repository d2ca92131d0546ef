"""mermaid_classifier_amd -- MI355X (gfx950) implementation of the PySpacer EfficientNet patch
feature-extraction path of data-mermaid/mermaid-classifier, behind the reference's own
extractor / predictor interfaces.  Importing the package is cheap; the HIP library is loaded on
first use and its absence is an error (there is no CPU fallback)."""

from .extractor import EfficientNetExtractor, build_extractor_class, resolve_device, verify_device_numerics  # noqa: F401
from .inference import ManifestError, Predictor, load_predictor, SCHEMA_VERSION, TASK_NAME  # noqa: F401
from .backbone import Backbone, crop_patches_device, FEATURE_DIM  # noqa: F401

__all__ = [
    "EfficientNetExtractor", "build_extractor_class", "resolve_device", "verify_device_numerics",
    "ManifestError", "Predictor", "load_predictor", "SCHEMA_VERSION", "TASK_NAME",
    "Backbone", "crop_patches_device", "FEATURE_DIM",
]
