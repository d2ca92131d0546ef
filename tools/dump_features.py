#!/usr/bin/env python3
"""GPU box: features of 24 fixed patches -> gpurun_out/<name>.npy (to compare builds bit for bit across gpurun calls)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mermaid_classifier_amd.backbone import Backbone
from mermaid_classifier_amd.synthetic import synthetic_state_dict
from oracle import efficientnet_b0_ref as ref
sd = synthetic_state_dict(0, dict(np.load(ROOT / "tests/golden/synth_bn_stats.npz")))
bb = Backbone(sd, device=0, max_batch=16)
p = np.concatenate([ref.natural_patches(16, seed=7), ref.synthetic_patches(8, seed=42)])
f = bb.extract(p)
out = ROOT / "gpurun_out" / (sys.argv[1] + ".npy")
np.save(out, f)
if len(sys.argv) > 2:
    prev = np.load(ROOT / "tools" / ("_" + sys.argv[2] + ".npy"))   # (gpurun_out/ does not travel: copy the baseline to tools/_<name>.npy)
    print("bitwise equal to", sys.argv[2], ":", np.array_equal(prev, f), " max|d|", np.abs(prev - f).max())
