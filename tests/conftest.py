import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth_sd():
    """Synthetic B0 weights (oracle generator + committed calibrated BN statistics)."""
    from oracle import efficientnet_b0_ref as bb
    stats = dict(np.load(GOLDEN / "synth_bn_stats.npz"))
    return bb.make_synthetic_state_dict(seed=0, bn_stats=stats)


@pytest.fixture(scope="session")
def synth_sd_b4():
    """Synthetic EfficientNet-B4 weights (BASELINE.json configs[4]) with their committed BN statistics."""
    from oracle import efficientnet_b0_ref as bb
    stats = {k: v.astype(np.float32) for k, v in np.load(GOLDEN / "synth_bn_stats_b4.npz").items()}
    return bb.make_synthetic_state_dict(seed=0, bn_stats=stats, arch="b4")


@pytest.fixture(scope="session")
def oracle_net(synth_sd):
    from oracle import efficientnet_b0_ref as bb
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    return bb.EfficientNetB0Ref(synth_sd)


@pytest.fixture(scope="session")
def checkpoint_path(synth_sd, tmp_path_factory):
    """efficientnet.pt in the pyspacer layout ({'net': {'module.<key>': tensor}})."""
    from oracle import efficientnet_b0_ref as bb
    p = tmp_path_factory.mktemp("weights") / "efficientnet.pt"
    bb.save_pyspacer_checkpoint(synth_sd, str(p))
    return p


@pytest.fixture(scope="session")
def golden_backbone():
    return dict(np.load(GOLDEN / "backbone_features.npz"))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1) + 1e-12)
