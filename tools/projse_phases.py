#!/usr/bin/env python3
"""In-kernel phase timing of proj_patch_kernel (per-tensor mode: MMC_KEEP_ACTIVATIONS=1 hands it a clock buffer):
cumulative shader cycles at the prologue's barriers, then the GEMM.  Development aid.

    python tools/projse_phases.py [batch]
"""
import os, sys
from pathlib import Path
import numpy as np
os.environ["MMC_KEEP_ACTIVATIONS"] = "1"
os.environ.setdefault("MMC_LANES", "1")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.synthetic import synthetic_state_dict
    sd = synthetic_state_dict(0, dict(np.load(ROOT / "tests/golden/synth_bn_stats.npz")))
    bb = Backbone(sd, device=0, max_batch=n)
    p = torch.from_numpy(np.random.default_rng(42).integers(0, 255, (n, 224, 224, 3), dtype=np.uint8)).cuda()
    f = torch.empty((n, 1280), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    for blk in range(3, 11):
        clk = bb.read_activation(f"b{blk}.clk", n * 8).reshape(n, 8)[:, :8]
        med = np.median(clk, axis=0)
        print(f"b{blk}.projse: prologue {med[0]:7.0f}  gemm {med[1]:7.0f} cycles (wave 0, last pair: k-loop {med[2]:7.0f}); "
              f"prologue barriers at pooled {med[3]:.0f}  fc1 (incl. its in-wave reduce) {med[4]:.0f}  fc2 {med[6]:.0f}  weights-in-LDS {med[7]:.0f}")

if __name__ == "__main__":
    main()
