#!/usr/bin/env python3
"""GPU box: B4 features under schedule switches, each in its own process (switches are read once): all variants must
meet the same bits where the arithmetic order is the same (thin_proj vs pw_gemm) and the golden gates otherwise."""
import os, subprocess, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from mermaid_classifier_amd.backbone import Backbone
from mermaid_classifier_amd.synthetic import synthetic_state_dict
from oracle import efficientnet_b0_ref as ref
stats = {k: v.astype(np.float32) for k, v in np.load(%r).items()}
bb = Backbone(synthetic_state_dict(0, stats, arch="b4"), device=0, max_batch=6)
np.save(sys.argv[1], bb.extract(np.concatenate([ref.natural_patches(3, seed=7), ref.synthetic_patches(3, seed=42)])))
''' % (str(ROOT), str(ROOT / "tests/golden/synth_bn_stats_b4.npz"))
outs = {}
with tempfile.TemporaryDirectory() as d:
    for tag, env in (("default", {}), ("thin0", {"MMC_THIN_PROJ": "0"}), ("projse0", {"MMC_PROJSE": "0"}), ("unfused", {"MMC_FUSE": "0"})):
        out = os.path.join(d, tag + ".npy")
        subprocess.run([sys.executable, "-c", CHILD, out], check=True, env={**os.environ, **env})
        outs[tag] = np.load(out)
g = np.load(ROOT / "tests/golden/backbone_b4_features.npz")
want = np.concatenate([g["natural4"][:3], g["noise4"][:3]])
for tag, f in outs.items():
    rel = np.linalg.norm(f - want, axis=1) / np.linalg.norm(want, axis=1)
    print(f"{tag:8s} rel-L2 vs oracle {np.array2string(rel, precision=2)}  bitwise == default: {np.array_equal(f, outs['default'])}")
assert np.array_equal(outs["default"], outs["thin0"]), "thin_proj must reproduce pw_gemm bit for bit"
print("OK")
