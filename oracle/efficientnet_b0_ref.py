"""CPU oracle for the EfficientNet-B0 backbone of the PySpacer extractor path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  The product path
(``mermaid_classifier_amd``) never does; it fails loudly when the HIP library is
missing.

PARITY UNPINNED for the backbone.  The arithmetic being restated lives in the
third-party dependency ``pyspacer==0.14.0`` (reference ``pyproject.toml:25``,
``uv.lock:2549-2566``), which is neither vendored in the reference nor installed
here, and the reference's own tests never run it
(``tests/pyspacer/test_build_feature_bucket.py:3-5``).  This file restates the
published algorithm of the network pyspacer vendors -- lukemelas
EfficientNet-PyTorch, model ``efficientnet-b0`` -- as a plain fp32 PyTorch
function over a state dict with that project's key names, anchored on the
reference call sites:

* ``scripts/build_feature_bucket.py:402-413`` -- weights loading order
  (``load_weights`` -> ``.to(device)`` -> ``.eval()``),
* ``scripts/build_feature_bucket.py:430-437`` -- ``net.extract_features(batch)``
  on a stacked ``(B, 3, 224, 224)`` fp32 tensor, result ``(B, 1280)``,
* ``scripts/build_feature_bucket.py:469-473`` -- the seed-42 random patches of the
  reference's only numeric gate for this path.

It is cross-checked in ``tests/test_oracle.py`` against an independent local
implementation of the same published network (HF ``transformers``
``EfficientNetModel`` built offline from a config, weights remapped), which is the
strongest pin available offline.

Published algorithm (EfficientNet-B0, eval mode):
  stem   conv3x3 s2 (TF "same" padding, no bias) -> BN(eps 1e-3) -> swish
  16 x MBConv: [expand 1x1 -> BN -> swish] (skipped when expand ratio is 1)
               depthwise kxk stride s (TF "same") -> BN -> swish
               squeeze-excite: global mean -> 1x1(+bias) -> swish -> 1x1(+bias)
                               -> sigmoid -> channel-wise multiply
               project 1x1 -> BN, (+ input when stride 1 and Cin == Cout)
  head   conv1x1 320->1280 -> BN -> swish -> global mean -> flatten
"""

from __future__ import annotations

import io
from typing import Dict, List, NamedTuple, Optional

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
FEATURE_DIM = 1280
CROP_SIZE = 224
NUM_FC_CLASSES = 1275  # pyspacer builds the net with num_classes=1275; _fc is unused by extract_features


class BlockArgs(NamedTuple):
    kernel: int
    stride: int
    expand: int
    cin: int
    cout: int


# Stage table of efficientnet-b0 (width 1.0, depth 1.0); 16 blocks.
_STAGES = [
    # repeats, kernel, stride, expand, cin, cout
    (1, 3, 1, 1, 32, 16),
    (2, 3, 2, 6, 16, 24),
    (2, 5, 2, 6, 24, 40),
    (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112),
    (4, 5, 2, 6, 112, 192),
    (1, 3, 1, 6, 192, 320),
]


def b0_blocks() -> List[BlockArgs]:
    blocks: List[BlockArgs] = []
    for rep, k, s, e, cin, cout in _STAGES:
        for r in range(rep):
            blocks.append(BlockArgs(k, s if r == 0 else 1, e, cin if r == 0 else cout, cout))
    return blocks


B0_BLOCKS = b0_blocks()
STEM_CH = 32
HEAD_IN = 320


class ArchRef(NamedTuple):
    """One member of the published family: B0's stage table under compound scaling
    (lukemelas ``round_filters`` / ``round_repeats``: widths to multiples of 8, never below
    90 % of the scaled value; repeats rounded up)."""
    name: str
    stem: int
    blocks: List[BlockArgs]
    head_in: int
    feature_dim: int


def _round_filters(c: int, width: float, divisor: int = 8) -> int:
    c = c * width
    new = max(divisor, int(c + divisor / 2) // divisor * divisor)
    if new < 0.9 * c:
        new += divisor
    return int(new)


def arch_ref(name: str = "b0") -> ArchRef:
    """``b0`` (the reference path's network) or ``b4`` (width 1.4, depth 1.8; BASELINE.json configs[4],
    not present in the reference -- same restatement, wider/deeper table)."""
    width, depth = {"b0": (1.0, 1.0), "b4": (1.4, 1.8)}[name]
    blocks: List[BlockArgs] = []
    for rep, k, s, e, cin, cout in _STAGES:
        cin, cout = _round_filters(cin, width), _round_filters(cout, width)
        for r in range(int(np.ceil(depth * rep))):
            blocks.append(BlockArgs(k, s if r == 0 else 1, e, cin if r == 0 else cout, cout))
    return ArchRef(name, _round_filters(32, width), blocks, blocks[-1].cout, _round_filters(1280, width))


assert arch_ref("b0").blocks == B0_BLOCKS


def se_channels(cin: int) -> int:
    return max(1, int(cin * 0.25))


def same_pad(size: int, k: int, s: int):
    """TF-style "same" padding of the static-image-size conv: (before, after)."""
    out = -(-size // s)
    pad = max((out - 1) * s + k - size, 0)
    return pad // 2, pad - pad // 2


# --------------------------------------------------------------------------
# Synthetic weights (there is no real efficientnet.pt offline).
# --------------------------------------------------------------------------

def expected_keys(arch: str = "b0") -> Dict[str, tuple]:
    """State-dict keys (without the ``module.`` prefix) and shapes of pyspacer's B0."""
    A = arch_ref(arch)
    keys: Dict[str, tuple] = {}

    def bn(prefix: str, c: int):
        keys[prefix + ".weight"] = (c,)
        keys[prefix + ".bias"] = (c,)
        keys[prefix + ".running_mean"] = (c,)
        keys[prefix + ".running_var"] = (c,)
        keys[prefix + ".num_batches_tracked"] = ()

    keys["_conv_stem.weight"] = (A.stem, 3, 3, 3)
    bn("_bn0", A.stem)
    for i, b in enumerate(A.blocks):
        p = f"_blocks.{i}."
        ce = b.cin * b.expand
        if b.expand != 1:
            keys[p + "_expand_conv.weight"] = (ce, b.cin, 1, 1)
            bn(p + "_bn0", ce)
        keys[p + "_depthwise_conv.weight"] = (ce, 1, b.kernel, b.kernel)
        bn(p + "_bn1", ce)
        cs = se_channels(b.cin)
        keys[p + "_se_reduce.weight"] = (cs, ce, 1, 1)
        keys[p + "_se_reduce.bias"] = (cs,)
        keys[p + "_se_expand.weight"] = (ce, cs, 1, 1)
        keys[p + "_se_expand.bias"] = (ce,)
        keys[p + "_project_conv.weight"] = (b.cout, ce, 1, 1)
        bn(p + "_bn2", b.cout)
    keys["_conv_head.weight"] = (A.feature_dim, A.head_in, 1, 1)
    bn("_bn1", A.feature_dim)
    keys["_fc.weight"] = (NUM_FC_CLASSES, A.feature_dim)
    keys["_fc.bias"] = (NUM_FC_CLASSES,)
    return keys


def synthetic_patches(n: int, seed: int = 42) -> np.ndarray:
    """(n,224,224,3) u8 patches drawn like the reference's numerics check
    (scripts/build_feature_bucket.py:469-473: default_rng(42).integers(0,255,...))."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, 255, (CROP_SIZE, CROP_SIZE, 3), dtype=np.uint8) for _ in range(n)])


def natural_patches(n: int, seed: int = 7) -> np.ndarray:
    """Image-like u8 patches: 1/f^alpha random fields (natural-image power spectrum),
    a shared luminance field plus weaker chroma fields, random mean and contrast."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, CROP_SIZE, CROP_SIZE, 3), dtype=np.uint8)
    fy = np.fft.fftfreq(CROP_SIZE)[:, None]
    fx = np.fft.fftfreq(CROP_SIZE)[None, :]
    rad = np.sqrt(fx * fx + fy * fy)
    rad[0, 0] = 1.0

    def field(alpha):
        spec = (rng.normal(size=rad.shape) + 1j * rng.normal(size=rad.shape)) / rad ** alpha
        spec[0, 0] = 0
        f = np.real(np.fft.ifft2(spec))
        return f / (f.std() + 1e-12)

    for i in range(n):
        alpha = rng.uniform(0.7, 1.8)
        lum = field(alpha)
        img = np.empty((CROP_SIZE, CROP_SIZE, 3))
        base = rng.uniform(70, 180, 3)
        contrast = rng.uniform(15, 60)
        for c in range(3):
            img[..., c] = base[c] + contrast * (lum + rng.uniform(0.1, 0.5) * field(alpha + 0.3))
        out[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def calibration_patches() -> np.ndarray:
    """Mixture the synthetic BN statistics are calibrated on: image-like fields plus
    white noise, so both kinds of test patch stay in the calibrated regime."""
    return np.concatenate([natural_patches(24, seed=99), synthetic_patches(8, seed=1234)])


def make_synthetic_state_dict(seed: int = 0, bn_stats: Optional[Dict[str, np.ndarray]] = None,
                              calib_patches: Optional[np.ndarray] = None, arch: str = "b0") -> Dict[str, torch.Tensor]:
    """Seeded synthetic B0 weights in the lukemelas/pyspacer key layout.

    The random part comes from mermaid_classifier_amd.synthetic (numpy default_rng(seed),
    stable across machines); this function adds the calibration.  BN running statistics
    are *calibrated*: like a trained net, running_mean/var track the statistics of the
    layer's input on a fixed calibration mixture, so activations stay O(1)
    through all 16 blocks instead of exploding or dying.  The calibrated statistics
    are data (tests/golden/synth_bn_stats.npz) so that every machine builds exactly
    the same weights; pass ``bn_stats=None`` to recompute them with ``calib_patches``.
    """
    from mermaid_classifier_amd.synthetic import synthetic_state_dict  # one generator for product and oracle
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_state_dict(seed, bn_stats, arch=arch).items()}
    if bn_stats is not None:
        return sd
    # ---- calibrate BN running stats on data, layer by layer --------------------
    if calib_patches is None:
        calib_patches = calibration_patches()
    x = transformation(calib_patches)

    def calib(pre: torch.Tensor, prefix: str):
        sd[prefix + ".running_mean"] = pre.mean(dim=(0, 2, 3)).clone()
        sd[prefix + ".running_var"] = pre.var(dim=(0, 2, 3), unbiased=False).clone() + 1e-4

    with torch.no_grad():
        _forward(sd, x, calibrate=calib, blocks=arch_ref(arch).blocks)
    return sd


def bn_stats_of(sd: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
    return {k: v.numpy() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}


def save_pyspacer_checkpoint(sd: Dict[str, torch.Tensor], path_or_buf) -> None:
    """Write weights the way pyspacer's efficientnet.pt is laid out [RECALL, SURVEY 8b]:
    ``torch.save({'net': DataParallel state dict})`` i.e. keys prefixed ``module.``."""
    torch.save({"net": {"module." + k: v for k, v in sd.items()}}, path_or_buf)


def checkpoint_bytes(sd: Dict[str, torch.Tensor]) -> bytes:
    buf = io.BytesIO()
    save_pyspacer_checkpoint(sd, buf)
    return buf.getvalue()


# --------------------------------------------------------------------------
# transformation(): ToTensor + Normalize(ImageNet mean/std)  [RECALL R6]
# --------------------------------------------------------------------------

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def transformation(patches_u8: np.ndarray) -> torch.Tensor:
    """(N,224,224,3) u8 HWC -> (N,3,224,224) fp32, exactly torchvision's
    ``Compose([ToTensor(), Normalize(mean, std)])``: ``(x/255 - mean)/std``
    evaluated in fp32 in that order (call site scripts/build_feature_bucket.py:420-432)."""
    x = torch.from_numpy(np.ascontiguousarray(patches_u8)).permute(0, 3, 1, 2).to(torch.float32).div(255)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - mean) / std


# --------------------------------------------------------------------------
# The network
# --------------------------------------------------------------------------

def _swish(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def _bn(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=BN_EPS)


def _conv_same(x: torch.Tensor, w: torch.Tensor, stride: int, groups: int = 1, bias=None) -> torch.Tensor:
    k = w.shape[-1]
    ph = same_pad(x.shape[2], k, stride)
    pw = same_pad(x.shape[3], k, stride)
    if any(ph) or any(pw):
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
    return F.conv2d(x, w, bias, stride=stride, groups=groups)


def _q(x: torch.Tensor, emulate_fp16: bool) -> torch.Tensor:
    return x.half().float() if emulate_fp16 else x


def _forward(sd, x: torch.Tensor, calibrate=None, taps: Optional[dict] = None,
             emulate_fp16: bool = False, blocks: Optional[List[BlockArgs]] = None) -> torch.Tensor:
    """extract_features: (B,3,224,224) fp32 -> (B,1280) fp32 (``blocks`` None = B0's table).

    ``calibrate(pre_bn_tensor, bn_prefix)`` is the weight generator's hook;
    ``taps`` collects named intermediates (NCHW fp32) for per-kernel parity tests;
    ``emulate_fp16`` rounds every tensor the HIP path stores in HBM to fp16 (a
    tolerance study aid, not the oracle's definition)."""
    def bn(prefix, t):
        if calibrate is not None:
            calibrate(t, prefix)
        return _bn(sd, prefix, t)

    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().clone()

    x = _q(x, emulate_fp16)
    x = _swish(bn("_bn0", _conv_same(x, sd["_conv_stem.weight"], 2)))
    x = _q(x, emulate_fp16)
    tap("stem", x)
    for i, b in enumerate(B0_BLOCKS if blocks is None else blocks):
        p = f"_blocks.{i}."
        inp = x
        ce = b.cin * b.expand
        if b.expand != 1:
            x = _swish(bn(p + "_bn0", _conv_same(x, sd[p + "_expand_conv.weight"], 1)))
            x = _q(x, emulate_fp16)
            tap(f"b{i}.expand", x)
        x = _swish(bn(p + "_bn1", _conv_same(x, sd[p + "_depthwise_conv.weight"], b.stride, groups=ce)))
        pooled = x.mean(dim=(2, 3), keepdim=True)  # pooled from the fp32 result
        x = _q(x, emulate_fp16)
        tap(f"b{i}.dw", x)
        s = _swish(F.conv2d(pooled, sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"]))
        s = F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"])
        gate = torch.sigmoid(s)
        tap(f"b{i}.gate", gate)
        x = _q(gate * x, emulate_fp16)  # gate stays fp32; one rounding of the product
        x = bn(p + "_bn2", _conv_same(x, sd[p + "_project_conv.weight"], 1))
        if b.stride == 1 and b.cin == b.cout:
            x = x + inp
        x = _q(x, emulate_fp16)
        tap(f"b{i}.out", x)
    x = _swish(bn("_bn1", _conv_same(x, sd["_conv_head.weight"], 1)))
    x = x.mean(dim=(2, 3))
    tap("features", x)
    return x


class EfficientNetB0Ref:
    """Oracle net: ``load_weights(stream)`` + ``extract_features(batch)`` like pyspacer's."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], arch: str = "b0"):
        self.arch = arch_ref(arch)
        want = expected_keys(arch)
        missing = sorted(set(want) - set(state_dict))
        unexpected = sorted(set(state_dict) - set(want))
        if missing or unexpected:
            raise KeyError(f"state dict mismatch: missing={missing[:8]} unexpected={unexpected[:8]}")
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v for k, v in state_dict.items()}

    @classmethod
    def load_weights(cls, stream) -> "EfficientNetB0Ref":
        """[RECALL R1] ``torch.load(stream)['net']`` with the ``module.`` prefix stripped."""
        ckpt = torch.load(stream, map_location="cpu", weights_only=True)
        net = ckpt["net"]
        return cls({k[7:] if k.startswith("module.") else k: v for k, v in net.items()})

    @torch.no_grad()
    def extract_features(self, batch: torch.Tensor, taps: Optional[dict] = None,
                         emulate_fp16: bool = False) -> torch.Tensor:
        return _forward(self.sd, batch.to(torch.float32), taps=taps, emulate_fp16=emulate_fp16, blocks=self.arch.blocks)


def patches_to_features(net: EfficientNetB0Ref, patches_u8: np.ndarray, batch_size: int = 10) -> np.ndarray:
    """Restates _DeviceCachingExtractor.patches_to_features on CPU
    (scripts/build_feature_bucket.py:415-446): batches of ``batch_size`` (pyspacer
    default 10), transform -> stack -> extract_features -> rows of python floats."""
    n = len(patches_u8)
    out = []
    for b in range(int(np.ceil(n / batch_size))):
        chunk = patches_u8[b * batch_size:(b + 1) * batch_size]
        out.append(net.extract_features(transformation(np.asarray(chunk))).numpy())
    if not out:
        return np.zeros((0, net.arch.feature_dim), dtype=np.float32)
    return np.concatenate(out).astype(np.float32)
