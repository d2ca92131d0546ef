#!/usr/bin/env python3
"""What fp8 (e4m3, per-tensor activation scale, per-output-channel weight scale) operands on the pointwise convs would cost
in feature accuracy -- CPU study with the oracle (BASELINE.json configs[4] asks for fp8 weights/activations).
Result on the synthetic weights (image-like patches): rel-L2 ~5e-2, cosine ~0.9985 for B0 and B4 alike -- 50x over the
north-star L2 gate (1e-3) and under the reference's own cosine gate (0.999, scripts/build_feature_bucket.py:456-457)."""
import sys
from pathlib import Path
import numpy as np
import torch
import torch.nn.functional as F
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import efficientnet_b0_ref as ref


def q8(x):
    s = x.abs().amax().clamp(min=1e-12) / 448.0
    return (x / s).to(torch.float8_e4m3fn).float() * s


def q8c(w):
    s = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-12) / 448.0
    return (w / s).to(torch.float8_e4m3fn).float() * s


def forward(net, x, fp8):
    sd = net.sd
    conv1 = (lambda t, w: F.conv2d(q8(t), q8c(w))) if fp8 else (lambda t, w: F.conv2d(t, w))
    x = ref._swish(ref._bn(sd, "_bn0", ref._conv_same(x, sd["_conv_stem.weight"], 2)))
    for i, b in enumerate(net.arch.blocks):
        p, inp, ce = f"_blocks.{i}.", x, b.cin * b.expand
        if b.expand != 1:
            x = ref._swish(ref._bn(sd, p + "_bn0", conv1(x, sd[p + "_expand_conv.weight"])))
        x = ref._swish(ref._bn(sd, p + "_bn1", ref._conv_same(x, sd[p + "_depthwise_conv.weight"], b.stride, groups=ce)))
        s = ref._swish(F.conv2d(x.mean(dim=(2, 3), keepdim=True), sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"]))
        g = torch.sigmoid(F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"]))
        x = ref._bn(sd, p + "_bn2", conv1(g * x, sd[p + "_project_conv.weight"]))
        if b.stride == 1 and b.cin == b.cout:
            x = x + inp
    return ref._swish(ref._bn(sd, "_bn1", conv1(x, sd["_conv_head.weight"]))).mean(dim=(2, 3))


for arch, f in (("b0", "synth_bn_stats.npz"), ("b4", "synth_bn_stats_b4.npz")):
    stats = {k: v.astype(np.float32) for k, v in np.load(ROOT / "tests/golden" / f).items()}
    net = ref.EfficientNetB0Ref(ref.make_synthetic_state_dict(seed=0, bn_stats=stats, arch=arch), arch=arch)
    x = ref.transformation(ref.natural_patches(4, seed=7))
    with torch.no_grad():
        a, b = forward(net, x, False).numpy(), forward(net, x, True).numpy()
    r = np.linalg.norm(a - b, axis=1) / np.linalg.norm(a, axis=1)
    c = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
    print(arch, "fp8 pointwise operands: rel-L2", r, "cosine", c)
