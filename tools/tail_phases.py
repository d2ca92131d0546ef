#!/usr/bin/env python3
"""In-kernel phase timing of the PRODUCTION tail7 launch (block 11 .. features in one kernel, both lanes running):
MMC_TAIL_CLK=1 makes the library hand tail7_kernel a clock buffer [patch][section][phase] (shader cycles, workgroup
thread 0).  Sections: b11 (front half: expand + depthwise stride 2 | SE | gate | project), b12..b15 (expand | dw | fc1 |
fc2 | gate | project), head, whole kernel.  Development aid.

    python tools/tail_phases.py [batch]
"""
import os, sys
from pathlib import Path
import numpy as np
os.environ["MMC_TAIL_CLK"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.synthetic import synthetic_state_dict
    sd = synthetic_state_dict(0, dict(np.load(ROOT / "tests/golden/synth_bn_stats.npz")))
    bb = Backbone(sd, device=0, max_batch=n)
    p = torch.from_numpy(np.random.default_rng(42).integers(0, 255, (n, 224, 224, 3), dtype=np.uint8)).cuda()
    f = torch.empty((n, 1280), dtype=torch.float32, device="cuda")
    for _ in range(5):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    clk = bb.read_activation("tail.clk", n * 64).reshape(n, 8, 8)
    med = np.median(clk, axis=0)
    print("tune", [os.environ.get(f"MMC_T7_TUNE{i}", "0") for i in range(4)])
    print("b11 : " + "  ".join(f"{nm} {c:7.0f}" for nm, c in zip(["front", "se", "gate", "project"], med[0][:4])) + f"   total {med[0][:4].sum():8.0f}")
    names = ["expand", "dw", "fc1", "fc2", "gate", "project"]
    for s in range(1, 5):
        print(f"b{11 + s} : " + "  ".join(f"{nm} {c:7.0f}" for nm, c in zip(names, med[s][:6])) + f"   total {med[s][:6].sum():8.0f}")
    if med[7].any():
        print("b12 group loop (DW4), sums over 9 groups: expand MFMAs | SiLU + store | depthwise MFMAs | epilogue")
        print("  wave 0: " + " | ".join(f"{c:7.0f}" for c in med[7][:4]) + "    wave 4: " + " | ".join(f"{c:7.0f}" for c in med[7][4:]))
    if os.environ.get("MMC_MID14M") != "1":
        print(f"head: {med[5][0]:7.0f}   whole kernel {med[6][0]:8.0f} cycles (median over {n} workgroups)")
        return
    mc = bb.read_activation("mid14.clk", n * 128).reshape(n, 8, 16)[:, :2, :]       # 2 workgroups per patch (mid14m)
    mm = np.median(mc.reshape(-1, 16), axis=0)
    print("mid14m (block 10), per wave: staging + barrier | first group: expand | first group: depthwise | remaining groups")
    for nm, o in (("wave 0", 0), ("wave 4", 4), ("wave 7", 8)):
        print(f"  {nm}: {mm[o]:7.0f} | {mm[o + 1]:7.0f} | {mm[o + 2]:7.0f} | {mm[o + 3]:7.0f}")
    print(f"head: {med[5][0]:7.0f}   whole kernel {med[6][0]:8.0f} cycles (median over {n} workgroups)")

if __name__ == "__main__":
    main()
