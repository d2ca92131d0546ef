#!/usr/bin/env python3
"""BASELINE.json configs[4] on the GPU box: EfficientNet-B4 forward, batch 512 resident 224x224 u8 patches (fp16 storage /
MFMA operands, generic per-layer schedule).  Prints patches/s and the per-kernel HIP-event time of one pass.
Development aid; bench.py (configs[1], B0) is the judged entry point."""
import argparse, sys, time
from collections import defaultdict
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp8"])
    a = ap.parse_args()
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.synthetic import synthetic_state_dict
    stats = {k: v.astype(np.float32) for k, v in np.load(ROOT / "tests/golden/synth_bn_stats_b4.npz").items()}
    bb = Backbone(synthetic_state_dict(0, stats, arch="b4"), device=0, max_batch=a.batch, precision=a.precision)
    p = torch.from_numpy(np.random.default_rng(42).integers(0, 255, (a.batch, 224, 224, 3), dtype=np.uint8)).cuda()
    f = torch.empty((a.batch, bb.feature_dim), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"B4 {a.precision} batch {a.batch}: {dt*1e3:.2f} ms/pass -> {a.batch/dt:.0f} patches/s ({a.batch/dt*3.0e9/1e12:.1f} TFLOP/s at 3.00 GFLOP/patch); "
          f"workspace {bb.workspace_bytes/2**30:.2f} GiB, lanes {bb.lanes}")
    per = defaultdict(float)
    prof = bb.profile(p, f)
    top = sorted(prof, key=lambda kv: -kv[1])[:14]
    print("  slowest launches (one lane of", a.batch // bb.lanes, "patches):", ", ".join(f"{n.split('|')[0]}={ms*1e3:.0f}us" for n, ms in top))
    for name, ms in prof:
        kern = name.split("|")[1]
        kind = name.split("|")[0].split(".")[-1]
        per[f"{kind}:{kern.split('<')[0]}"] += ms
    tot = sum(per.values())
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v:8.3f} ms  {100*v/tot:5.1f} %")


if __name__ == "__main__":
    main()
