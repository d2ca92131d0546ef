#!/usr/bin/env python3
"""What fp8 (OCP e4m3) MFMA operands on the pointwise convs would cost in feature accuracy (BASELINE.json configs[4]
asks for an EfficientNet-B4 with fp8 weights/activations) -- TEST-SIDE STUDY on the CPU oracle, never shipped.

Operands are quantised exactly as an fp8 MFMA path would see them; accumulation, depthwise convs, squeeze-excite,
the residual stream and every stored tensor stay as in the fp32 oracle, so the numbers isolate the operand format.

  weights      per-output-channel scale (amax / 448), e4m3
  activations  per-tensor (r01 study) | per-patch | per-pixel (one scale per GEMM row, the finest an MFMA epilogue can
               undo with one multiply per output row)
  where        every pointwise conv | only the project convs | only the expand convs | only blocks from the 14x14 stage on

Gates: reference cosine >= 0.999 (scripts/build_feature_bucket.py:456-457), north-star rel-L2 < 1e-3.

    python tests/study_fp8.py > profiles/r02_fp8_study.txt
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import efficientnet_b0_ref as ref  # noqa: E402

E4M3_MAX = 448.0


def q_act(x, mode):
    """x: (B, C, H, W).  Scale granularity: tensor | patch | pixel."""
    if mode == "tensor":
        s = x.abs().amax()
    elif mode == "patch":
        s = x.abs().amax(dim=(1, 2, 3), keepdim=True)
    else:
        s = x.abs().amax(dim=1, keepdim=True)
    s = s.clamp(min=1e-12) / E4M3_MAX
    return (x / s).to(torch.float8_e4m3fn).float() * s


def q_w(w):
    s = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-12) / E4M3_MAX
    return (w / s).to(torch.float8_e4m3fn).float() * s


def forward(net, x, act_mode, where):
    """where(kind, block_index, H) -> bool: is this pointwise conv run on fp8 operands?"""
    sd = net.sd

    def conv1(t, w, kind, i):
        if act_mode is not None and where(kind, i, t.shape[-1]):
            return F.conv2d(q_act(t, act_mode), q_w(w))
        return F.conv2d(t, w)

    x = ref._swish(ref._bn(sd, "_bn0", ref._conv_same(x, sd["_conv_stem.weight"], 2)))
    for i, b in enumerate(net.arch.blocks):
        p, inp, ce = f"_blocks.{i}.", x, b.cin * b.expand
        if b.expand != 1:
            x = ref._swish(ref._bn(sd, p + "_bn0", conv1(x, sd[p + "_expand_conv.weight"], "expand", i)))
        x = ref._swish(ref._bn(sd, p + "_bn1", ref._conv_same(x, sd[p + "_depthwise_conv.weight"], b.stride, groups=ce)))
        s = ref._swish(F.conv2d(x.mean(dim=(2, 3), keepdim=True), sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"]))
        g = torch.sigmoid(F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"]))
        x = ref._bn(sd, p + "_bn2", conv1(g * x, sd[p + "_project_conv.weight"], "project", i))
        if b.stride == 1 and b.cin == b.cout:
            x = x + inp
    return ref._swish(ref._bn(sd, "_bn1", conv1(x, sd["_conv_head.weight"], "head", len(net.arch.blocks)))).mean(dim=(2, 3))


WHERE = {
    "every pointwise conv": lambda kind, i, h: True,
    "project convs only": lambda kind, i, h: kind == "project",
    "expand convs + head only": lambda kind, i, h: kind != "project",
    "14x14 and 7x7 stages only": lambda kind, i, h: h <= 14,
    "7x7 stage + head only": lambda kind, i, h: h <= 7,
}


def main():
    torch.set_num_threads(8)
    for arch, f in (("b0", "synth_bn_stats.npz"), ("b4", "synth_bn_stats_b4.npz")):
        stats = {k: v.astype(np.float32) for k, v in np.load(ROOT / "tests/golden" / f).items()}
        net = ref.EfficientNetB0Ref(ref.make_synthetic_state_dict(seed=0, bn_stats=stats, arch=arch), arch=arch)
        for pname, patches in (("image-like (1/f, seed 7)", ref.natural_patches(4, seed=7)),
                               ("white noise (seed 42)", ref.synthetic_patches(4, seed=42))):
            x = ref.transformation(patches)
            with torch.no_grad():
                base = forward(net, x, None, None).numpy()
                emu = net.extract_features(x, emulate_fp16=True).numpy()
            e = np.linalg.norm(emu - base, axis=1) / np.linalg.norm(base, axis=1)
            print(f"== EfficientNet-{arch.upper()}, {pname}, 4 patches; fp16 storage emulation for scale: rel-L2 max {e.max():.2e}")
            print(f"{'fp8 e4m3 operands on':<28}{'activation scale':<18}{'rel-L2 max':>12}{'mean':>11}{'cos min':>10}")
            for wname, where in WHERE.items():
                for mode in ("tensor", "patch", "pixel"):
                    with torch.no_grad():
                        y = forward(net, x, mode, where).numpy()
                    r = np.linalg.norm(y - base, axis=1) / np.linalg.norm(base, axis=1)
                    c = (y * base).sum(1) / np.linalg.norm(y, axis=1) / np.linalg.norm(base, axis=1)
                    print(f"{wname:<28}{mode:<18}{r.max():12.2e}{r.mean():11.2e}{c.min():10.5f}")
            print()


if __name__ == "__main__":
    main()
