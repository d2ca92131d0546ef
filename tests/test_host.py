"""CPU tests of the host logic: BN/normalisation folding, blob layout, checkpoint validation,
the C-ABI surface (symbols only -- no compute without a GPU), shims and loader validation."""

import ctypes
import json
import re
import struct
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ROOT, rel_l2


def _folded_forward(tensors, patches_u8, arch="b0"):
    """Evaluate the network from the FOLDED tensors (what the library consumes) in fp32 torch:
    proves fold() + the stem's u8-128 / padval trick reproduce the oracle's arithmetic."""
    from mermaid_classifier_amd.weights import get_arch
    A = get_arch(arch)
    t = {k: torch.from_numpy(v) for k, v in tensors}
    u = torch.from_numpy(patches_u8.astype(np.float32) - 128.0).permute(0, 3, 1, 2)   # (B,3,224,224)
    pv = t["stem.padval"].view(1, 3, 1, 1)
    up = pv.expand(u.shape[0], 3, 225, 225).clone()
    up[:, :, :224, :224] = u                                                          # pad right/bottom with padval
    w = t["stem.weight"].view(A.stem, 3, 3, 3).permute(0, 3, 1, 2)                    # (n,ky,kx,c)->(n,c,ky,kx)
    x = F.conv2d(up, w, t["stem.bias"], stride=2)
    x = x * torch.sigmoid(x)
    for i, (k, s, e, cin, cout) in enumerate(A.blocks):
        inp = x
        ce = cin * e
        if e != 1:
            x = F.conv2d(x, t[f"b{i}.expand.weight"].view(ce, cin, 1, 1), t[f"b{i}.expand.bias"])
            x = x * torch.sigmoid(x)
        h = x.shape[2]
        out = -(-h // s)
        pad = max((out - 1) * s + k - h, 0)
        x = F.pad(x, (pad // 2, pad - pad // 2, pad // 2, pad - pad // 2))
        x = F.conv2d(x, t[f"b{i}.dw.weight"].view(ce, 1, k, k), t[f"b{i}.dw.bias"], stride=s, groups=ce)
        x = x * torch.sigmoid(x)
        pooled = x.mean(dim=(2, 3))
        r = pooled @ t[f"b{i}.se.reduce.weight"].T + t[f"b{i}.se.reduce.bias"]
        r = r * torch.sigmoid(r)
        g = torch.sigmoid(r @ t[f"b{i}.se.expand.weight"].T + t[f"b{i}.se.expand.bias"])
        x = x * g[:, :, None, None]
        x = F.conv2d(x, t[f"b{i}.project.weight"].view(cout, ce, 1, 1), t[f"b{i}.project.bias"])
        if s == 1 and cin == cout:
            x = x + inp
    x = F.conv2d(x, t["head.weight"].view(A.feature_dim, A.head_in, 1, 1), t["head.bias"])
    x = x * torch.sigmoid(x)
    return x.mean(dim=(2, 3)).numpy()


def test_fold_reproduces_oracle(synth_sd, oracle_net):
    from mermaid_classifier_amd import weights
    from oracle import efficientnet_b0_ref as ref
    sd = {k: np.asarray(v.numpy(), np.float64) for k, v in synth_sd.items() if k in weights.expected_shapes()}
    tensors = weights.fold(sd)
    patches = np.concatenate([ref.synthetic_patches(1, seed=42), ref.natural_patches(1, seed=7)])
    with torch.no_grad():
        got = _folded_forward(tensors, patches)
        want = oracle_net.extract_features(ref.transformation(patches)).numpy()
    assert rel_l2(got, want).max() < 2e-5


def test_b4_arch_table_fold_and_blob(synth_sd_b4):
    """BASELINE.json configs[4]: the compound-scaled table (product and oracle derive it independently), its folding, and
    the blob header's arch id."""
    from mermaid_classifier_amd import weights
    from oracle import efficientnet_b0_ref as ref
    A = weights.get_arch("b4")
    R = ref.arch_ref("b4")
    assert (A.stem, A.head_in, A.feature_dim, len(A.blocks)) == (48, 448, 1792, 32) == (R.stem, R.head_in, R.feature_dim, len(R.blocks))
    assert [tuple(b) for b in R.blocks] == A.blocks
    assert weights.detect_arch(synth_sd_b4) is A
    sd = {k: np.asarray(v.numpy(), np.float64) for k, v in synth_sd_b4.items() if k in weights.expected_shapes(A)}
    tensors = weights.fold(sd, A)
    patches = ref.natural_patches(1, seed=7)
    with torch.no_grad():
        got = _folded_forward(tensors, patches, "b4")
        want = ref.EfficientNetB0Ref(synth_sd_b4, arch="b4").extract_features(ref.transformation(patches)).numpy()
    assert rel_l2(got, want).max() < 2e-5
    blob = weights.pack_backbone(sd, A)
    assert struct.unpack_from("<III", blob, 4) == (1, 1, len(tensors))
    with pytest.raises(ValueError):
        weights.get_arch("b7")


def test_blob_layout_roundtrip(synth_sd):
    from mermaid_classifier_amd import weights
    sd = {k: np.asarray(v.numpy(), np.float64) for k, v in synth_sd.items() if k in weights.expected_shapes()}
    blob = weights.pack_backbone(sd)
    assert blob[:4] == b"MMCW"
    version, arch, n = struct.unpack_from("<III", blob, 4)
    tensors = weights.fold(sd)
    assert (version, arch, n) == (1, 0, len(tensors)) and n == 3 + 16 * 10 - 2 + 2
    for i, (_, a) in enumerate(tensors):
        off, nb = struct.unpack_from("<QQ", blob, 16 + 16 * i)
        assert off % 256 == 0 and nb == a.nbytes
        assert np.array_equal(np.frombuffer(blob, np.float32, a.size, off).reshape(a.shape), a)


def test_checkpoint_validation_is_loud(synth_sd, checkpoint_path, tmp_path):
    from mermaid_classifier_amd import weights
    sd = weights.load_checkpoint(str(checkpoint_path))     # pyspacer layout: {'net': {'module.x': ...}}
    assert set(sd) == set(weights.expected_shapes())
    broken = {("module." + k): v for k, v in synth_sd.items() if "_blocks.3._se_reduce" not in k}
    broken["module._blocks.99.bogus"] = torch.zeros(1)
    p = tmp_path / "broken.pt"
    torch.save({"net": broken}, p)
    with pytest.raises(weights.WeightsError) as ei:
        weights.load_checkpoint(str(p))
    assert "_blocks.3._se_reduce.weight" in str(ei.value) and "_blocks.99.bogus" in str(ei.value)
    torch.save({"something": 1}, p)
    with pytest.raises(weights.WeightsError):
        weights.load_checkpoint(str(p))


def test_abi_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly the entry points include/mmc.h declares."""
    from mermaid_classifier_amd import _lib
    header = (ROOT / "include" / "mmc.h").read_text()
    declared = sorted(set(re.findall(r"\b(mmc_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(_lib.SYMBOLS)
    lib = _lib.lib()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.mmc_version() == 1
    assert lib.mmc_device_count() >= 0


def test_no_cpu_fallback_without_device(checkpoint_path):
    """Without a HIP device the product path refuses loudly instead of computing on the CPU."""
    from mermaid_classifier_amd import _lib
    if _lib.lib().mmc_device_count() > 0:
        pytest.skip("a HIP device is present")
    from mermaid_classifier_amd.backbone import Backbone
    with pytest.raises(RuntimeError, match="no HIP device"):
        Backbone(str(checkpoint_path), device=0, max_batch=4)
    from mermaid_classifier_amd import load_predictor
    with pytest.raises(RuntimeError, match="no HIP device"):
        load_predictor(GOLDEN / "head_fixture" / "model.pt", GOLDEN / "head_fixture" / "model.json")


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    from mermaid_classifier_amd import _lib
    monkeypatch.setenv("MMC_LIBRARY", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LibraryMissingError):
        _lib._load()


def test_product_package_never_imports_the_oracle():
    src = ROOT / "mermaid_classifier_amd"
    for f in src.rglob("*.py"):
        txt = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
    for f in (src / "csrc").iterdir():
        if not f.is_file():
            continue   # (csrc/_obj: the build's object files)
        assert "oracle" not in f.read_text(errors="ignore").lower().replace("oracle/efficientnet_b0_ref.py", ""), f


def test_params_from_torchscript_match_reference_parameters():
    from mermaid_classifier_amd.inference import params_from_torchscript
    io = np.load(GOLDEN / "head_fixture_io.npz")
    prm = params_from_torchscript(torch.jit.load(str(GOLDEN / "head_fixture" / "model.pt")))
    assert prm.dims == [8, 16, 5]
    for i in range(int(io["n_layers"])):
        assert np.array_equal(prm.weights[i], io[f"W{i}"]) and np.array_equal(prm.biases[i], io[f"b{i}"])
    assert np.array_equal(prm.a, io["a"]) and np.array_equal(prm.b, io["b"])
    big = params_from_torchscript(torch.jit.load(str(GOLDEN / "head108" / "model.pt")))
    assert big.dims == [1280, 500, 300, 100, 108]


def test_loader_manifest_errors_before_touching_the_gpu(tmp_path):
    """Mirrors reference tests/pyspacer/test_portable_artifact.py:122-147."""
    from mermaid_classifier_amd import ManifestError, load_predictor
    pt = GOLDEN / "head_fixture" / "model.pt"
    manifest = json.loads((GOLDEN / "head_fixture" / "model.json").read_text())
    for mutate, pattern in (
        (lambda m: m.update(schema_version=2), "schema_version"),
        (lambda m: m.update(classes=m["classes"][:-1]), "class-count"),
        (lambda m: m.update(input_dim=9), "input_dim"),
    ):
        m = json.loads(json.dumps(manifest))
        mutate(m)
        j = tmp_path / "model.json"
        j.write_text(json.dumps(m))
        with pytest.raises(ManifestError, match=pattern):
            load_predictor(pt, j)


def test_shim_image_features_roundtrip(tmp_path):
    from mermaid_classifier_amd.spacer_shim import DataLocation, ImageFeatures, PointFeatures
    pfs = [PointFeatures(3, 4, [0.5, 1.5]), PointFeatures(7, 1, [2.0, -1.0])]
    feats = ImageFeatures(pfs, True, 2, 2)
    assert np.array_equal(feats.get_array((7, 1)), [2.0, -1.0])
    loc = DataLocation("filesystem", str(tmp_path / "i1.featurevector"))
    feats.store(loc)
    back = ImageFeatures.load(loc)
    assert back.serialize() == feats.serialize()
    with pytest.raises(ValueError):
        DataLocation("s3", "k")


def test_extractor_constructor_shapes(checkpoint_path):
    """Both reference call shapes construct without touching the GPU (net is built lazily)."""
    from mermaid_classifier_amd import EfficientNetExtractor, build_extractor_class
    from mermaid_classifier_amd.spacer_shim import DataLocation
    loc = DataLocation("filesystem", str(checkpoint_path))
    a = build_extractor_class()(data_locations={"weights": loc}, device="cuda", batch_size=10)
    b = EfficientNetExtractor(data_locations={"weights": loc})
    assert a.feature_dim == b.feature_dim == 1280 and a.CROP_SIZE == 224 and a.DATA_LOCATION_KEYS == ["weights"]
    ds, remote = a.load_datastream("weights")
    assert remote is False and len(ds.read()) > 1_000_000
    with pytest.raises(ValueError):
        build_extractor_class()(data_locations={}, device="cuda", batch_size=10)
    with pytest.raises(ValueError):
        build_extractor_class()(data_locations={"weights": loc}, device="cuda", batch_size=0)


def test_extract_reference_features_cli_stacks_in_file_then_point_order(tmp_path):
    """Reference scripts/extract_reference_features.py:50-59 semantics."""
    from mermaid_classifier_amd.extract_reference_features import main, stack_feature_files
    from mermaid_classifier_amd.spacer_shim import DataLocation, ImageFeatures, PointFeatures
    files = []
    for i, rows in enumerate(([[1, 2, 3], [4, 5, 6]], [[7, 8, 9]])):
        pfs = [PointFeatures(r, r, [float(v) for v in row]) for r, row in enumerate(rows)]
        p = tmp_path / f"i{i}.featurevector"
        ImageFeatures(pfs, True, 3, len(pfs)).store(DataLocation("filesystem", str(p)))
        files.append(str(p))
    x = stack_feature_files(files)
    assert x.dtype == np.float32 and x.tolist() == [[1, 2, 3], [4, 5, 6], [7, 8, 9]]
    out = tmp_path / "ref.npy"
    main(["--out", str(out)] + files)
    assert np.array_equal(np.load(out), x)


def test_check_extract_inputs():
    from mermaid_classifier_amd.pipeline import check_extract_inputs
    img = np.zeros((300, 400, 3), np.uint8)
    check_extract_inputs(img, [(0, 0), (299, 399)])
    with pytest.raises(ValueError, match="outside"):
        check_extract_inputs(img, [(300, 0)])
    with pytest.raises(ValueError, match="exceed"):
        check_extract_inputs(np.zeros((200, 400, 3), np.uint8), [(1, 1)])


def test_numerics_gate_runs_as_the_reference_calls_it(checkpoint_path, oracle_net, caplog):
    """scripts/build_feature_bucket.py:860-861 calls ``verify_device_numerics(extractor, weights_loc, batch_size, device)``
    and the gate builds its CPU side with ``cls(..., device="cpu")`` from the replaced class factory (:475-480).  With a
    stand-in pyspacer installed, that whole sequence runs unmodified: the cpu-side instance takes pyspacer's own torch path
    (load_weights -> to(cpu) -> eval -> transformation -> extract_features), the gate logs and returns None, raises on a
    mismatch, and is a no-op for device == "cpu".  (The device side is a stub here: no GPU in the CPU suite; the -m gpu test
    test_numerics_gate_against_stock_cpu_path runs the same sequence against the HIP extractor.)"""
    import logging
    import fake_spacer
    from mermaid_classifier_amd import build_extractor_class, verify_device_numerics
    from mermaid_classifier_amd.spacer_shim import DataLocation
    from oracle import efficientnet_b0_ref as ref
    loc = DataLocation("filesystem", str(checkpoint_path))
    with fake_spacer.installed() as Base:
        cls = build_extractor_class()
        assert issubclass(cls, Base)
        cpu_ex = cls(data_locations={"weights": loc}, device="cpu", batch_size=3)
        patches = ref.synthetic_patches(4, seed=42)
        feats, remote = cpu_ex.patches_to_features(list(patches))
        assert remote is False and cpu_ex._cached_net.device == "cpu" and cpu_ex._cached_net.evaluated
        want = ref.patches_to_features(oracle_net, patches)
        np.testing.assert_allclose(np.asarray(feats), want, rtol=0, atol=5e-5)   # the CPU arithmetic in batches of 3 (conv summation order varies with the batch)
        assert cpu_ex.patches_to_features(list(patches[:1]))[1] is False            # cached net: loaded_remote only once

        class DeviceStub:   # stands for the HIP extractor: features a hair away from / far away from the CPU side
            def __init__(self, scramble):
                self.scramble = scramble

            def patches_to_features(self, ps):
                f = ref.patches_to_features(oracle_net, np.stack([np.asarray(p) for p in ps]))
                f = f * (1 + 1e-4 * np.sin(np.arange(f.shape[1]))).astype(np.float32)
                return (f[:, ::-1] if self.scramble else f).tolist(), False

        with caplog.at_level(logging.INFO, logger="mermaid_classifier_amd.extractor"):
            assert verify_device_numerics(DeviceStub(False), loc, 4, "cuda") is None      # positional, as the reference calls it
        assert "Device numerics check (cuda vs cpu, 8 random patches): min_cos=" in caplog.text
        with pytest.raises(RuntimeError, match="Device numerics check FAILED on cuda"):
            verify_device_numerics(DeviceStub(True), loc, 4, "cuda")
        assert verify_device_numerics(None, None, 4, "cpu") is None                       # :459-460
    # without pyspacer there is no CPU side to build: loud, not a fallback
    cls = build_extractor_class()
    with pytest.raises(RuntimeError, match="needs pyspacer"):
        cls(data_locations={"weights": loc}, device="cpu", batch_size=3)
    with pytest.raises(RuntimeError, match="needs pyspacer"):
        verify_device_numerics(object(), loc, 4, "cuda")


def test_golden_torchscript_archives_carry_no_reference_source_text():
    """tests/golden/*/model.pt are fixtures (the artifact contract itself).  torch.jit.save also writes `*.debug_pkl` members
    holding the scripted module's Python source text (the reference's inference/head.py) and its path; make_golden.py strips
    them (strip_debug_pkl) -- reference source must not travel.  The archives still load (tests that use them do)."""
    import zipfile
    from pathlib import Path
    pts = sorted((Path(__file__).resolve().parent / "golden").glob("*/model.pt"))
    assert len(pts) >= 2
    for pt in pts:
        z = zipfile.ZipFile(pt)
        names = z.namelist()
        assert not [n for n in names if n.endswith(".debug_pkl")], pt
        for n in names:
            assert b"/root/reference" not in z.read(n), (pt, n)


def test_fp8_e4m3_encoder_matches_an_independent_implementation():
    """include/mmc.h mmc_fp8_e4m3_encode (the host-side quantiser of the fp8 project weights, BASELINE configs[4]) against
    torch's float8_e4m3fn cast: round to nearest even, subnormals, saturation at +-448, zero below half the smallest subnormal."""
    import torch
    from mermaid_classifier_amd import _lib
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 100, 100000), rng.normal(0, 1, 100000), rng.normal(0, 0.01, 50000),
                        np.array([0, 448, 449, 463.9, 464, 480, 1e9, -448, -500, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 2.0 ** -6,
                                  0.0145, -0.0, 17.0, 18.0, 19.0])]).astype(np.float32)
    out = np.zeros(len(x), np.uint8)
    _lib.check(_lib.lib().mmc_fp8_e4m3_encode(x.ctypes.data, out.ctypes.data, len(x)))
    want = torch.from_numpy(x).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(out, want)


def test_build_recompiles_only_stale_translation_units(tmp_path, monkeypatch):
    """build.py's staleness rule: an object is stale when its source or one of the headers it includes is newer; the library
    when any source or header is.  (Pure host logic on temporary files: hipcc is not run.)"""
    import os
    import time
    from mermaid_classifier_amd import build as B
    csrc = tmp_path / "pkg" / "csrc"
    (tmp_path / "include").mkdir()
    csrc.mkdir(parents=True)
    for src, deps in B.SOURCES.items():
        for f in [src] + deps:
            p = (csrc / f)
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text("")
    obj = csrc / "_obj"
    obj.mkdir()
    monkeypatch.setattr(B, "CSRC", csrc)
    monkeypatch.setattr(B, "OBJ", obj)
    monkeypatch.setattr(B, "OUT", tmp_path / "pkg" / "lib.so")
    assert B.needs_build() and all(B._stale(s) for s in B.SOURCES)
    now = time.time()
    for s in B.SOURCES:
        (obj / (s + ".o")).write_text("")
        os.utime(obj / (s + ".o"), (now + 10, now + 10))
    B.OUT.write_text("")
    os.utime(B.OUT, (now + 20, now + 20))
    assert not B.needs_build() and not any(B._stale(s) for s in B.SOURCES)
    os.utime(csrc / "k_tail.hip", (now + 30, now + 30))            # one kernel file edited: only its object is stale
    assert B.needs_build() and [s for s in B.SOURCES if B._stale(s)] == ["k_tail.hip"]
    os.utime(csrc / "k_tail.hip", (now, now))
    os.utime(csrc / "device_common.h", (now + 30, now + 30))       # the shared header: the five kernel units, nothing else
    assert sorted(s for s in B.SOURCES if B._stale(s)) == ["k_early.hip", "k_generic.hip", "k_mbconv.hip", "k_mid.hip", "k_tail.hip"]


def test_header_is_plain_c99(tmp_path):
    """include/mmc.h is what a C (or cgo / JNI / FFI) host includes: it must compile as C99, warnings as errors."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text(f'#include "{ROOT / "include" / "mmc.h"}"\n'
                   "int main(void) { unsigned char id[MMC_DIST_ID_BYTES]; mmc_dist* d = 0; (void)id; (void)d; return mmc_version() == 0; }\n")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
