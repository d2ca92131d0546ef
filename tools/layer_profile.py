#!/usr/bin/env python3
"""Per-launch table on the GPU box: HIP-event ms, algorithmic GB/s and TFLOP/s of every launch of one
256-patch pass (median of --passes).  Development aid; bench.py is the judged entry point."""
import argparse, os, sys
os.environ.setdefault("MMC_PROFILE_SERIAL", "1")   # isolated kernels: one lane at a time
from collections import defaultdict
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--passes", type=int, default=7)
    a = ap.parse_args()
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.synthetic import synthetic_state_dict
    from mermaid_classifier_amd import schedule
    sd = synthetic_state_dict(0, dict(np.load(ROOT / "tests/golden/synth_bn_stats.npz")))
    bb = Backbone(sd, device=0, max_batch=a.batch)
    p = torch.from_numpy(np.random.default_rng(42).integers(0, 255, (a.batch, 224, 224, 3), dtype=np.uint8)).cuda()
    f = torch.empty((a.batch, 1280), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    t = defaultdict(list)
    order = []
    for _ in range(a.passes):
        seen = defaultdict(int)
        for name, ms in bb.profile(p, f):
            seen[name] += 1
            key = name if seen[name] == 1 else None
            if key is None:
                t[name][-1] += 0.0  # second lane of the same layer: keep the first lane's timing only
                continue
            if name not in t:
                order.append(name)
            t[name].append(ms)
    sub = -(-a.batch // bb.lanes)
    alg = {l.name: l for l in schedule.b0_launches(sub)}
    print(f'lanes={bb.lanes} patches/launch={sub}')
    tot = 0.0
    print(f"{'launch':14s} {'kernel':26s} {'us':>8s} {'MB':>8s} {'GB/s':>8s} {'TF/s':>7s}")
    for name in order:
        layer, kern = name.split("|")
        ms = float(np.median(t[name]))
        tot += ms
        l = alg[layer]
        print(f"{layer:14s} {kern:26s} {ms*1e3:8.1f} {l.bytes/1e6:8.1f} {l.bytes/ms/1e6:8.0f} {l.flops/ms/1e9:7.1f}")
    print(f"sum of launches (one lane of {sub}): {tot:.3f} ms  -> {sub/tot*1e3:.0f} patches/s (event-timed, serialised)")
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"back-to-back: {dt*1e3:.3f} ms/pass -> {a.batch/dt:.0f} patches/s")

if __name__ == "__main__":
    main()
