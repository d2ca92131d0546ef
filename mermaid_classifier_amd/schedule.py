"""Algorithmic work of every launch of the EfficientNet-B0 schedule (per batch of B patches).

Used by bench.py for the roofline object and quoted in DESIGN.md.  "Algorithmic bytes" are the
compulsory HBM bytes of the launch as scheduled: every input tensor read once, every output
written once, weights once -- re-reads through caches are NOT counted (they show up as PMC
``traffic`` above this figure).  Names match the launch names of ``mmc_backbone_profile``.
"""

from __future__ import annotations

from typing import Dict, List, NamedTuple

from .weights import B0_BLOCKS, FEATURE_DIM


class Launch(NamedTuple):
    name: str
    kind: str          # stem | expand | dw | se | project | head
    bytes: int         # algorithmic HBM bytes
    flops: int         # 2*MAC (MFMA/VALU useful work)


def b0_launches(batch: int) -> List[Launch]:
    out: List[Launch] = []
    h = 112
    out.append(Launch("stem", "stem", batch * (224 * 224 * 3 + h * h * 32 * 2) + 32 * 32 * 2,
                      2 * batch * h * h * 32 * 27))
    # fused alternative of stem + block-0 depthwise (stem_dw_kernel): the stem tensor never reaches HBM
    out.append(Launch("stem+b0.dw", "stem_dw", batch * (224 * 224 * 3 + h * h * 32 * 2) + 32 * 32 * 2 + 9 * 32 * 4,
                      2 * batch * h * h * 32 * (27 + 9)))
    for i, (k, s, e, cin, cout) in enumerate(B0_BLOCKS):
        ce = cin * e
        cs = max(1, cin // 4)
        ho = -(-h // s)
        m_in, m_out = batch * h * h, batch * ho * ho
        if e != 1:
            out.append(Launch(f"b{i}.expand", "expand", m_in * (cin + ce) * 2 + ce * cin * 2, 2 * m_in * cin * ce))
            # fused alternative (mbconv_a_kernel): the expanded tensor never reaches HBM
            out.append(Launch(f"b{i}.mbconv", "mbconv", m_in * cin * 2 + m_out * ce * 2 + ce * cin * 2 + k * k * ce * 4,
                              2 * m_in * cin * ce + 2 * m_out * ce * k * k))
        out.append(Launch(f"b{i}.dw", "dw", (m_in + m_out) * ce * 2 + k * k * ce * 4, 2 * m_out * ce * k * k))
        out.append(Launch(f"b{i}.gate", "se", batch * ce * 4 * 2 + 2 * cs * ce * 4, 2 * batch * 2 * cs * ce))
        res = m_out * cout * 2 if (s == 1 and cin == cout) else 0
        out.append(Launch(f"b{i}.project", "project", m_out * (ce + cout) * 2 + res + batch * ce * 4 + cout * ce * 2,
                          2 * m_out * ce * cout))
        if i == 1:   # block 1's fused kernel with block 0's SE scale + project conv folded in (reads b0's depthwise output)
            # (block 0's depthwise output is counted ONCE, like every input: what the kernel re-reads per channel chunk
            # shows up as PMC traffic above this figure)
            out.append(Launch("b0.project+b1.mbconv", "mbconv", m_in * 32 * 2 + m_out * ce * 2 + ce * cin * 2 + k * k * ce * 4,
                              2 * m_in * cin * ce + 2 * m_out * ce * k * k + 2 * 2 * m_in * 32 * 16))
        if 3 <= i <= 10:   # squeeze-excite + project in one launch (proj_patch_kernel): no gate tensor in HBM
            out.append(Launch(f"b{i}.projse", "projse", m_out * (ce + cout) * 2 + res + batch * ce * 4 + cout * ce * 2 + 2 * cs * ce * 2,
                              2 * m_out * ce * cout + 2 * batch * 2 * cs * ce))
        h = ho
    # blocks 12..15 chained in one launch (tail7_kernel): only the 7x7 block input/output and the weights move
    tail = [l for l in out if l.name.split(".")[0] in ("b12", "b13", "b14", "b15") and l.kind in ("mbconv", "se", "project")]
    w_bytes = 4 * (1152 * 192 * 2 + 15 * 1152 * 4 + 2 * 48 * 1152 * 2) + (3 * 192 + 320) * 1152 * 2
    out.append(Launch("b12-15.tail", "tail", batch * 49 * (192 + 320) * 2 + w_bytes, sum(l.flops for l in tail)))
    # ... and with block 11's squeeze-excite + project in front and the head conv behind: one launch from the last
    # 14x14 depthwise output to the feature vector
    b11 = [l for l in out if l.name in ("b11.gate", "b11.project")]
    head_flops = 2 * batch * 49 * 320 * FEATURE_DIM
    out.append(Launch("b11-head.tail", "tail", batch * (49 * 672 * 2 + 672 * 4 + FEATURE_DIM * 4) + w_bytes
                      + 192 * 672 * 2 + 2 * 28 * 672 * 2 + FEATURE_DIM * 320 * 2,
                      sum(l.flops for l in tail) + sum(l.flops for l in b11) + head_flops))
    # ... and with block 11's front half (expand + depthwise stride 2) inside as well: from block 10's output to the features
    b11f = [l for l in out if l.name == "b11.mbconv"]
    out.append(Launch("b11all-head.tail", "tail", batch * (196 * 112 * 2 + FEATURE_DIM * 4) + w_bytes
                      + 672 * 112 * 2 + 15 * 672 * 4 + 192 * 672 * 2 + 2 * 28 * 672 * 2 + FEATURE_DIM * 320 * 2,
                      sum(l.flops for l in tail) + sum(l.flops for l in b11) + sum(l.flops for l in b11f) + head_flops))
    out.append(Launch("head", "head", batch * h * h * 320 * 2 + batch * FEATURE_DIM * 4 + FEATURE_DIM * 320 * 2,
                      2 * batch * h * h * 320 * FEATURE_DIM))
    return out


def totals(batch: int, launched=None) -> Dict[str, float]:
    """Totals over the launches actually issued (names from mmc_backbone_profile); default = unfused."""
    ls = b0_launches(batch)
    if launched is not None:
        names = set(launched)
        ls = [l for l in ls if l.name in names]
    else:
        ls = [l for l in ls if l.kind not in ("mbconv", "stem_dw", "tail", "projse")]
    return {"bytes": float(sum(l.bytes for l in ls)), "flops": float(sum(l.flops for l in ls)),
            "bytes_per_patch": sum(l.bytes for l in ls) / batch, "flops_per_patch": sum(l.flops for l in ls) / batch}


# MI355X peaks used to normalise (from /opt/skills/guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0          # 8 TB/s spec (6.3 TB/s achievable by a streaming copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16
