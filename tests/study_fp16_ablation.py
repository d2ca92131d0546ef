#!/usr/bin/env python3
"""Which fp16 roundings cost what: ablation of the HIP path's storage/operand precision on the oracle.

TEST-SIDE STUDY (imports oracle/, never shipped): the fp32 oracle forward is re-run with ONE class of
tensor rounded to fp16 at a time, then with all classes at once, then with all-but-one -- on the
reference's own gate inputs (seed-42 white noise, scripts/build_feature_bucket.py:469-473) and on
image-like patches.  Classes (what the HIP kernels round):

  w_pw     1x1-conv weights (expand / project / head), BN folded in, as fp16 MFMA operands
  w_dw     depthwise taps, BN folded in, fp16 (v_dot2c operands)
  stem     stem output (only in HBM for the unfused schedule; LDS fp16 in stem_dw)
  expand   expanded tensor (LDS, fp16)
  dw       depthwise output (HBM fp16; pool sums are taken before rounding)
  gated    gate * depthwise output (the project conv's MFMA operand)
  out      block output = the residual stream (HBM fp16)
  out_op   block output rounded ONLY as the next expand's MFMA operand; the residual stream itself
           stays fp32 (the proposed fix: b<i>.out in fp32 in HBM)

    python tests/study_fp16_ablation.py [n_patches] > profiles/r02_fp16_ablation.txt
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import efficientnet_b0_ref as ref  # noqa: E402

CLASSES = ["w_pw", "w_dw", "stem", "expand", "dw", "gated", "out", "out_op"]


def h(t):
    return t.half().float()


def fold(sd, conv, bnp):
    """conv weight with BN scale folded in, and the BN shift as bias (what weights.py packs)."""
    w = sd[conv]
    s = sd[bnp + ".weight"] / torch.sqrt(sd[bnp + ".running_var"] + ref.BN_EPS)
    b = sd[bnp + ".bias"] - sd[bnp + ".running_mean"] * s
    return w * s.view(-1, 1, 1, 1), b


def forward(net, x, on):
    sd = net.sd
    q = lambda name, t: h(t) if name in on else t  # noqa: E731
    w, b = fold(sd, "_conv_stem.weight", "_bn0")
    x = q("stem", ref._swish(ref._conv_same(x, w, 2, bias=b)))
    for i, blk in enumerate(net.arch.blocks):
        p, ce = f"_blocks.{i}.", blk.cin * blk.expand
        inp = x                                   # residual stream
        xin = h(x) if "out_op" in on else x       # MFMA operand view of it
        if blk.expand != 1:
            w, b = fold(sd, p + "_expand_conv.weight", p + "_bn0")
            xin = q("expand", ref._swish(F.conv2d(xin, q("w_pw", w), b)))
        w, b = fold(sd, p + "_depthwise_conv.weight", p + "_bn1")
        d = ref._swish(ref._conv_same(xin, q("w_dw", w), blk.stride, groups=ce, bias=b))
        pooled = d.mean(dim=(2, 3), keepdim=True)
        d = q("dw", d)
        s = ref._swish(F.conv2d(pooled, sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"]))
        g = torch.sigmoid(F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"]))
        w, b = fold(sd, p + "_project_conv.weight", p + "_bn2")
        x = F.conv2d(q("gated", g * d), q("w_pw", w), b)
        if blk.stride == 1 and blk.cin == blk.cout:
            x = x + inp
        x = q("out", x)
    w, b = fold(sd, "_conv_head.weight", "_bn1")
    xin = h(x) if "out_op" in on else x
    return ref._swish(F.conv2d(xin, q("w_pw", w), b)).mean(dim=(2, 3))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    stats = {k: v.astype(np.float32) for k, v in np.load(ROOT / "tests/golden/synth_bn_stats.npz").items()}
    net = ref.EfficientNetB0Ref(ref.make_synthetic_state_dict(seed=0, bn_stats=stats))
    sets = {"noise (seed 42, the reference's gate inputs)": ref.synthetic_patches(n, seed=42),
            "image-like (1/f fields, seed 7)": ref.natural_patches(n, seed=7)}
    torch.set_num_threads(8)
    for title, patches in sets.items():
        x = ref.transformation(patches)
        with torch.no_grad():
            base = forward(net, x, set()).numpy()
            chk = net.extract_features(x).numpy()
        print(f"== {title}, {n} patches; folded-BN fp32 forward vs oracle: rel-L2 "
              f"{(np.linalg.norm(base - chk, axis=1) / np.linalg.norm(chk, axis=1)).max():.2e}")

        def run(on):
            with torch.no_grad():
                y = forward(net, x, set(on)).numpy()
            r = np.linalg.norm(y - base, axis=1) / np.linalg.norm(base, axis=1)
            return r.max(), r.mean()

        print(f"{'rounded to fp16':<44}{'max rel-L2':>12}{'mean':>12}")
        for c in CLASSES:
            mx, mn = run([c])
            print(f"only {c:<39}{mx:12.2e}{mn:12.2e}")
        hip = ["w_pw", "w_dw", "stem", "expand", "dw", "gated", "out"]
        mx, mn = run(hip)
        print(f"{'all (the r01 HIP path)':<44}{mx:12.2e}{mn:12.2e}")
        for c in hip:
            mx, mn = run([k for k in hip if k != c])
            print(f"all but {c:<36}{mx:12.2e}{mn:12.2e}")
        fix = ["w_pw", "w_dw", "stem", "expand", "dw", "gated", "out_op"]
        mx, mn = run(fix)
        print(f"{'all, residual stream fp32 (out -> out_op)':<44}{mx:12.2e}{mn:12.2e}")
        mx, mn = run([k for k in fix if k not in ("w_pw",)])
        print(f"{'  ... and fp32 pointwise weights':<44}{mx:12.2e}{mn:12.2e}")
        mx, mn = run([k for k in fix if k not in ("expand",)])
        print(f"{'  ... and fp32 expanded tensor':<44}{mx:12.2e}{mn:12.2e}")
        mx, mn = run([k for k in fix if k not in ("dw", "gated")])
        print(f"{'  ... and fp32 dw output + gated operand':<44}{mx:12.2e}{mn:12.2e}")
        print()


if __name__ == "__main__":
    main()
