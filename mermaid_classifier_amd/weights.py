"""efficientnet.pt -> folded fp32 tensors -> the weights blob of ``mmc_backbone_create``.

Replaces the host half of ``EfficientNetExtractor.load_weights(stream)`` as used at
reference ``scripts/build_feature_bucket.py:402-413``: read the checkpoint, validate
its keys loudly, then (new) fold every eval-mode BatchNorm into the preceding
convolution and fold ``transformation()`` (ToTensor + Normalize, call site
``scripts/build_feature_bucket.py:420-423``) into the stem.

Checkpoint layout [RECALL R1-R3, SURVEY 8b]: ``torch.save({'net': state_dict})`` whose
keys carry a ``module.`` (DataParallel) prefix; lukemelas EfficientNet-PyTorch names;
``_fc`` (1275 x 1280) is present and unused by ``extract_features``.

Blob (all tensors fp32, natural layouts; the library does its own MFMA packing):
  header  magic "MMCW", u32 version=1, u32 arch (0 = B0, 1 = B4), u32 n_tensors
  table   n_tensors x (u64 offset, u64 nbytes), offsets 256-B aligned
  order   stem.weight [Cstem][27] (ky,kx,c fastest; BN scale and 1/(255 std) folded)
          stem.bias [Cstem]   (BN shift + folded mean term)
          stem.padval [3]  (255*mean - 128: the u8-128 value that normalises to 0)
          per block i:
            [expand.weight [Ce][Cin], expand.bias [Ce]]       (absent when expand ratio is 1)
            dw.weight [Ce][k][k], dw.bias [Ce]
            se.reduce.weight [Cs][Ce], se.reduce.bias [Cs], se.expand.weight [Ce][Cs], se.expand.bias [Ce]
            project.weight [Cout][Ce], project.bias [Cout]
          head.weight [F][Chead_in], head.bias [F]          (B0: 1280 x 320, B4: 1792 x 448)

Architectures: ``b0`` is the network pyspacer's EfficientNetExtractor builds (the reference path);
``b4`` (width 1.4, depth 1.8 of the same published family, still on 224x224 patches) is BASELINE.json
configs[4] -- it does not exist in the reference and runs on the library's generic per-layer kernels.
"""

from __future__ import annotations

import io
import struct
from typing import Dict, List, Tuple

import numpy as np

BN_EPS = 1e-3
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
FEATURE_DIM = 1280

# (kernel, stride, expand, cin, cout) for the 16 MBConv blocks of efficientnet-b0
B0_BLOCKS: List[Tuple[int, int, int, int, int]] = [
    (3, 1, 1, 32, 16), (3, 2, 6, 16, 24), (3, 1, 6, 24, 24), (5, 2, 6, 24, 40),
    (5, 1, 6, 40, 40), (3, 2, 6, 40, 80), (3, 1, 6, 80, 80), (3, 1, 6, 80, 80),
    (5, 1, 6, 80, 112), (5, 1, 6, 112, 112), (5, 1, 6, 112, 112), (5, 2, 6, 112, 192),
    (5, 1, 6, 192, 192), (5, 1, 6, 192, 192), (5, 1, 6, 192, 192), (3, 1, 6, 192, 320),
]

# The published EfficientNet family: B0's stage table scaled by (width, depth) coefficients.
# (repeats, kernel, stride, expand, cin, cout)
_B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
              (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


def _round_filters(c: int, width: float, divisor: int = 8) -> int:
    c = c * width
    new = max(divisor, int(c + divisor / 2) // divisor * divisor)
    if new < 0.9 * c:
        new += divisor
    return int(new)


class Arch:
    """Stem width, block list and feature width of one member of the family."""

    def __init__(self, name: str, arch_id: int, width: float, depth: float):
        self.name, self.arch_id = name, arch_id
        self.stem = _round_filters(32, width)
        self.blocks: List[Tuple[int, int, int, int, int]] = []
        for rep, k, s, e, cin, cout in _B0_STAGES:
            cin, cout = _round_filters(cin, width), _round_filters(cout, width)
            for r in range(int(np.ceil(depth * rep))):
                self.blocks.append((k, s if r == 0 else 1, e, cin if r == 0 else cout, cout))
        self.head_in = self.blocks[-1][4]
        self.feature_dim = _round_filters(1280, width)


ARCHS: Dict[str, Arch] = {"b0": Arch("b0", 0, 1.0, 1.0), "b4": Arch("b4", 1, 1.4, 1.8)}
assert ARCHS["b0"].blocks == B0_BLOCKS and ARCHS["b0"].feature_dim == FEATURE_DIM


def get_arch(arch) -> Arch:
    if isinstance(arch, Arch):
        return arch
    if arch in (None, 0):
        return ARCHS["b0"]
    if arch == 1:
        return ARCHS["b4"]
    try:
        return ARCHS[str(arch).lower().replace("efficientnet-", "")]
    except KeyError:
        raise ValueError(f"unknown architecture {arch!r} (known: {sorted(ARCHS)})") from None


def detect_arch(sd) -> Arch:
    """Which family member a state dict holds, from the stem width."""
    w = sd.get("_conv_stem.weight", sd.get("module._conv_stem.weight"))
    if w is None:
        raise WeightsError("state dict has no _conv_stem.weight")
    for a in ARCHS.values():
        if int(np.shape(w)[0]) == a.stem:
            return a
    raise WeightsError(f"stem width {np.shape(w)[0]} matches none of {sorted(ARCHS)}")


def expected_shapes(arch=None) -> Dict[str, tuple]:
    A = get_arch(arch)
    shapes: Dict[str, tuple] = {}

    def bn(prefix, c):
        for suffix in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{prefix}.{suffix}"] = (c,)

    shapes["_conv_stem.weight"] = (A.stem, 3, 3, 3)
    bn("_bn0", A.stem)
    for i, (k, s, e, cin, cout) in enumerate(A.blocks):
        p = f"_blocks.{i}."
        ce = cin * e
        cs = max(1, int(cin * 0.25))
        if e != 1:
            shapes[p + "_expand_conv.weight"] = (ce, cin, 1, 1)
            bn(p + "_bn0", ce)
        shapes[p + "_depthwise_conv.weight"] = (ce, 1, k, k)
        bn(p + "_bn1", ce)
        shapes[p + "_se_reduce.weight"] = (cs, ce, 1, 1)
        shapes[p + "_se_reduce.bias"] = (cs,)
        shapes[p + "_se_expand.weight"] = (ce, cs, 1, 1)
        shapes[p + "_se_expand.bias"] = (ce,)
        shapes[p + "_project_conv.weight"] = (cout, ce, 1, 1)
        bn(p + "_bn2", cout)
    shapes["_conv_head.weight"] = (A.feature_dim, A.head_in, 1, 1)
    bn("_bn1", A.feature_dim)
    return shapes


class WeightsError(ValueError):
    """The checkpoint does not look like pyspacer's efficientnet-b0 weights."""


def load_checkpoint(stream, arch=None) -> Dict[str, np.ndarray]:
    """Read an ``efficientnet.pt`` byte stream / path -> {key: fp64 ndarray} with the
    ``module.`` prefix stripped.  Fails loudly, listing unexpected / missing keys.
    ``arch`` None = efficientnet-b0, the network the reference path loads."""
    import torch

    if isinstance(stream, (bytes, bytearray)):
        stream = io.BytesIO(stream)
    ckpt = torch.load(stream, map_location="cpu", weights_only=True)
    if not isinstance(ckpt, dict):
        raise WeightsError(f"checkpoint is a {type(ckpt).__name__}, expected a dict with key 'net'")
    if "net" in ckpt and isinstance(ckpt["net"], dict):
        net = ckpt["net"]
    elif all(hasattr(v, "shape") for v in ckpt.values()):
        net = ckpt  # a bare state dict
    else:
        raise WeightsError(f"checkpoint has top-level keys {sorted(ckpt)[:6]}; expected 'net'")
    sd = {}
    for k, v in net.items():
        k = k[7:] if k.startswith("module.") else k
        sd[k] = v.detach().cpu().numpy()
    A = get_arch(arch)
    want = expected_shapes(A)
    ignorable = lambda k: k.endswith("num_batches_tracked") or k.startswith("_fc.")  # noqa: E731
    missing = sorted(k for k in want if k not in sd)
    unexpected = sorted(k for k in sd if k not in want and not ignorable(k))
    bad_shape = sorted(f"{k}: {tuple(sd[k].shape)} != {want[k]}" for k in want if k in sd and tuple(sd[k].shape) != want[k])
    if missing or unexpected or bad_shape:
        raise WeightsError(
            f"efficientnet-{A.name} checkpoint mismatch: "
            f"missing={missing[:10]}{'...' if len(missing) > 10 else ''} "
            f"unexpected={unexpected[:10]}{'...' if len(unexpected) > 10 else ''} "
            f"bad_shape={bad_shape[:10]}")
    return {k: np.asarray(sd[k], dtype=np.float64) for k in want}


def _bn_scale_shift(sd, prefix):
    scale = sd[prefix + ".weight"] / np.sqrt(sd[prefix + ".running_var"] + BN_EPS)
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return scale, shift


def fold(sd: Dict[str, np.ndarray], arch=None) -> List[Tuple[str, np.ndarray]]:
    """BN folding (fp64) -> ordered list of (name, fp32 array) in blob order."""
    A = get_arch(arch)
    out: List[Tuple[str, np.ndarray]] = []
    mean = np.asarray(IMAGENET_MEAN)
    std = np.asarray(IMAGENET_STD)
    # stem: y = g * sum w * ((x/255 - mean)/std) + h, with x = u + 128 (u = u8 - 128, exact in fp16)
    g, h = _bn_scale_shift(sd, "_bn0")
    w = sd["_conv_stem.weight"].transpose(0, 2, 3, 1)        # [n][ky][kx][c]
    s_c = 1.0 / (255.0 * std)
    t_c = (128.0 - 255.0 * mean) / (255.0 * std)
    out.append(("stem.weight", (w * s_c * g[:, None, None, None]).reshape(A.stem, 27)))
    out.append(("stem.bias", g * (w * t_c).sum(axis=(1, 2, 3)) + h))
    out.append(("stem.padval", 255.0 * mean - 128.0))
    for i, (k, s, e, cin, cout) in enumerate(A.blocks):
        p = f"_blocks.{i}."
        ce = cin * e
        if e != 1:
            g, h = _bn_scale_shift(sd, p + "_bn0")
            out.append((f"b{i}.expand.weight", sd[p + "_expand_conv.weight"].reshape(ce, cin) * g[:, None]))
            out.append((f"b{i}.expand.bias", h))
        g, h = _bn_scale_shift(sd, p + "_bn1")
        out.append((f"b{i}.dw.weight", sd[p + "_depthwise_conv.weight"].reshape(ce, k, k) * g[:, None, None]))
        out.append((f"b{i}.dw.bias", h))
        cs = max(1, int(cin * 0.25))
        out.append((f"b{i}.se.reduce.weight", sd[p + "_se_reduce.weight"].reshape(cs, ce)))
        out.append((f"b{i}.se.reduce.bias", sd[p + "_se_reduce.bias"]))
        out.append((f"b{i}.se.expand.weight", sd[p + "_se_expand.weight"].reshape(ce, cs)))
        out.append((f"b{i}.se.expand.bias", sd[p + "_se_expand.bias"]))
        g, h = _bn_scale_shift(sd, p + "_bn2")
        out.append((f"b{i}.project.weight", sd[p + "_project_conv.weight"].reshape(cout, ce) * g[:, None]))
        out.append((f"b{i}.project.bias", h))
    g, h = _bn_scale_shift(sd, "_bn1")
    out.append(("head.weight", sd["_conv_head.weight"].reshape(A.feature_dim, A.head_in) * g[:, None]))
    out.append(("head.bias", h))
    return [(n, np.ascontiguousarray(a, dtype=np.float32)) for n, a in out]


def pack_backbone(sd: Dict[str, np.ndarray], arch=None) -> bytes:
    A = get_arch(arch)
    tensors = fold(sd, A)
    n = len(tensors)
    table_end = 16 + 16 * n
    off = (table_end + 255) // 256 * 256
    entries = []
    for _, a in tensors:
        entries.append((off, a.nbytes))
        off = (off + a.nbytes + 255) // 256 * 256
    buf = bytearray(off)
    buf[0:16] = b"MMCW" + struct.pack("<III", 1, A.arch_id, n)
    for i, (o, nb) in enumerate(entries):
        struct.pack_into("<QQ", buf, 16 + 16 * i, o, nb)
    for (o, nb), (_, a) in zip(entries, tensors):
        buf[o:o + nb] = a.tobytes()
    return bytes(buf)


def pack_from_stream(stream, arch=None) -> bytes:
    return pack_backbone(load_checkpoint(stream, arch), arch)
