"""A stand-in for the slice of pyspacer the extractor seam touches, installed into sys.modules for the duration of a test
(pattern: the reference's own tests stub ``spacer.*`` the same way, tests/pyspacer/test_build_feature_bucket.py:127-137).

``spacer.extractors.EfficientNetExtractor``   constructor(data_locations), load_datastream, classmethod load_weights,
                                              __call__ (CPU crop + patches_to_features), BATCH_SIZE / CROP_SIZE
``spacer.extractors.torch_extractors.transformation``   ToTensor + Normalize

The network behind ``load_weights`` is the CPU oracle (tests only): with it the reference's whole
``verify_device_numerics`` call sequence runs unmodified against the HIP extractor.
"""

import contextlib
import io
import sys
import types

import numpy as np


class _OracleTorchNet:
    """What pyspacer's ``load_weights`` returns, as far as the reference uses it: .to(), .eval(), .extract_features()."""

    def __init__(self, net):
        self.net = net
        self.device = None
        self.evaluated = False

    def to(self, device):
        self.device = str(device)
        return self

    def eval(self):
        self.evaluated = True
        return self

    def extract_features(self, batch):
        return self.net.extract_features(batch)


@contextlib.contextmanager
def installed():
    from oracle import efficientnet_b0_ref as ref, pyspacer_ref
    from mermaid_classifier_amd import extractor as ex_mod
    from mermaid_classifier_amd import spacer_shim

    class EfficientNetExtractor:
        DATA_LOCATION_KEYS = ["weights"]
        CROP_SIZE = 224
        BATCH_SIZE = 10

        def __init__(self, data_locations, data_hashes=None):
            self.data_locations = data_locations
            self.data_hashes = data_hashes or {}

        def load_datastream(self, key):
            loc = self.data_locations[key]
            return io.BytesIO(spacer_shim.load_bytes(loc)), loc.storage_type in ("s3", "url")

        @classmethod
        def load_weights(cls, stream):
            return _OracleTorchNet(ref.EfficientNetB0Ref.load_weights(stream))

        @property
        def feature_dim(self):
            return 1280

        def __call__(self, im, rowcols):
            patches = pyspacer_ref.crop_patches(np.asarray(im), rowcols, 224)
            feats, remote = self.patches_to_features(list(patches))
            pfs = [spacer_shim.PointFeatures(int(r), int(c), list(f)) for (r, c), f in zip(rowcols, feats)]
            return spacer_shim.ImageFeatures(pfs, True, 1280, len(feats)), spacer_shim.ExtractFeaturesReturnMsg(remote, 0.0)

    def transformation():
        return lambda img: ref.transformation(np.asarray(img)[None])[0]

    spacer = types.ModuleType("spacer")
    extractors = types.ModuleType("spacer.extractors")
    torch_extractors = types.ModuleType("spacer.extractors.torch_extractors")
    extractors.EfficientNetExtractor = EfficientNetExtractor
    torch_extractors.transformation = transformation
    spacer.extractors = extractors
    extractors.torch_extractors = torch_extractors
    names = {"spacer": spacer, "spacer.extractors": extractors, "spacer.extractors.torch_extractors": torch_extractors}
    saved = {k: sys.modules.get(k) for k in names}
    saved_cls = dict(ex_mod._cls_cache)
    sys.modules.update(names)
    ex_mod._cls_cache.clear()
    try:
        yield EfficientNetExtractor
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        ex_mod._cls_cache.clear()
        ex_mod._cls_cache.update(saved_cls)
