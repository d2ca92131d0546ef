#!/usr/bin/env python3
"""Debug aid (GPU box): per-tensor comparison of the B4 generic schedule against the oracle's taps."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MMC_KEEP_ACTIVATIONS"] = "1"
from mermaid_classifier_amd.backbone import Backbone
from oracle import efficientnet_b0_ref as ref
stats = {k: v.astype(np.float32) for k, v in np.load(ROOT / "tests/golden/synth_bn_stats_b4.npz").items()}
sd = ref.make_synthetic_state_dict(seed=0, bn_stats=stats, arch="b4")
net = ref.EfficientNetB0Ref(sd, arch="b4")
patches = ref.natural_patches(4, seed=7)[2:4]
taps = {}
want = net.extract_features(ref.transformation(patches), taps=taps).numpy()
bb = Backbone({k: v.numpy() for k, v in sd.items()}, device=0, max_batch=2)
got = bb.extract(patches)
for name, t in taps.items():
    if name == "features":
        continue
    o = t.numpy()
    o = o.reshape(o.shape[0], o.shape[1]) if name.endswith(".gate") else o.transpose(0, 2, 3, 1)
    try:
        g = bb.read_activation(name, o.size).reshape(o.shape)
    except ValueError:
        continue   # fused schedule: the expanded tensor lives only in LDS
    err = np.sqrt(np.mean((g - o) ** 2, axis=tuple(range(1, o.ndim)))) / (np.sqrt(np.mean(o ** 2, axis=tuple(range(1, o.ndim)))) + 1e-12)
    print(f"{name:12s} rel_rms={err} max|o|={np.abs(o).max():.3e} max|g|={np.abs(g).max():.3e}")
print("features rel-L2", np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1))
