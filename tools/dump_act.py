#!/usr/bin/env python3
"""GPU box: one intermediate tensor (MMC_KEEP_ACTIVATIONS=1) of 2 fixed patches -> gpurun_out/<name>.npy; optional compare."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["MMC_KEEP_ACTIVATIONS"] = "1"
from mermaid_classifier_amd.backbone import Backbone
from mermaid_classifier_amd.synthetic import synthetic_state_dict
from oracle import efficientnet_b0_ref as ref
sd = synthetic_state_dict(0, dict(np.load(ROOT / "tests/golden/synth_bn_stats.npz")))
bb = Backbone(sd, device=0, max_batch=2)
bb.extract(ref.natural_patches(2, seed=7))
out = {}
for t, n in (("b5.out", 2 * 196 * 80), ("b6.dw", 2 * 196 * 480), ("b6.out", 2 * 196 * 80)):
    out[t] = bb.read_activation(t, n)
np.savez(ROOT / "gpurun_out" / (sys.argv[1] + ".npz"), **out)
if len(sys.argv) > 2:
    prev = np.load(ROOT / "tools" / ("_" + sys.argv[2] + ".npz"))
    for t in out:
        d = out[t] != prev[t]
        print(t, "differing elements", int(d.sum()), "of", d.size, " max|d|", float(np.abs(out[t] - prev[t]).max()))
