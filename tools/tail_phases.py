#!/usr/bin/env python3
"""In-kernel phase timing of tail7_kernel (debug build path: MMC_KEEP_ACTIVATIONS=1 runs blocks 12..14 one
launch each and records shader cycles per phase).  Development aid."""
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
    names = ["expand", "dw", "fc1", "fc2", "gate", "project"]
    for blk in (12, 13, 14, 15):
        clk = bb.read_activation(f"b{blk}.clk", n * 8).reshape(n, 8)[:, :6]
        med = np.median(clk, axis=0)
        print(f"b{blk}: " + "  ".join(f"{nm} {c:8.0f}" for nm, c in zip(names, med)) + f"   total {med.sum():8.0f} cycles")
    projse(bb, n)

def projse(bb, n):
    for blk in range(3, 11):
        clk = bb.read_activation(f"b{blk}.clk", n * 8).reshape(n, 8)[:, :8]
        med = np.median(clk, axis=0)
        print(f"b{blk}.projse: prologue {med[0]:8.0f}  gemm {med[1]:8.0f} cycles (wave 0, last pair: k-loop {med[2]:8.0f}); prologue barriers at " + " ".join(f"{v:.0f}" for v in med[3:8]))


if __name__ == "__main__":
    main()
