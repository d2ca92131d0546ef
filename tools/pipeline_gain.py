#!/usr/bin/env python3
"""GPU box: patches/s of one device-resident call of n patches (max_batch 256) -- n = 256 is bench.py's step, larger n
shows what the chunk pipelining inside a call buys the cross-image batching path (pipeline.BatchedExtractor, n = 1024)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from mermaid_classifier_amd.backbone import Backbone
from mermaid_classifier_amd.synthetic import synthetic_state_dict
sd = synthetic_state_dict(0, dict(np.load(ROOT / "tests/golden/synth_bn_stats.npz")))
bb = Backbone(sd, device=0, max_batch=256)
for n in (256, 512, 1024, 2048, 4096):
    p = torch.from_numpy(np.random.default_rng(42).integers(0, 255, (n, 224, 224, 3), dtype=np.uint8)).cuda()
    f = torch.empty((n, 1280), dtype=torch.float32, device="cuda")
    for _ in range(4):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    reps = max(3, 8192 // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        bb.extract(p, out=f)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(0, n, 256):
            bb.extract(p[i:i + 256], out=f[i:i + 256])
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / reps
    print(f"n={n:5d}: one call {n/dt:9.0f} patches/s ({dt*1e3:.2f} ms)   {n//256} calls of 256: {n/dt2:9.0f} patches/s")
