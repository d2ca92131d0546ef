"""-m gpu: end-to-end parity of the HIP path, called through the C ABI, against the CPU oracle
and the committed golden fixtures."""

from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, check_labels, cosine, rel_l2

pytestmark = pytest.mark.gpu

# Tolerances (fp16 storage/MFMA operands, fp32 accumulation; see DESIGN.md "Numerics"):
#   image-like patches: per-vector relative L2 < 1e-3   (north_star gate)
#   seed-42 white-noise patches (the reference's own gate inputs): cosine >= 0.999
#   (scripts/build_feature_bucket.py:456-457) and relative L2 < 9e-3 = 1.25 x 7.24e-3, the error the fp32 oracle itself
#   makes on these 8 patches when exactly the tensors the HIP path keeps in fp16 are rounded to fp16
#   (profiles/r02_fp16_ablation.txt, tests/study_fp16_ablation.py).  That table is also why 1e-3 is out of reach on white
#   noise with fp16 MFMA operands: every single class of rounding (pointwise weights 4.3e-3, depthwise taps 3.2e-3,
#   depthwise output 3.3e-3, gated operand 2.8e-3, block outputs 2.6e-3, expanded tensor 2.0e-3, stem 1.7e-3) is on
#   its own already above it, they add in quadrature, and keeping the residual stream in fp32 (still rounded as the next
#   expand's operand) changes 7.24e-3 to 6.93e-3.  On image-like patches the same roundings give 5.9e-4.
TOL_NATURAL = 1e-3
TOL_NOISE = 9e-3
COS_GATE = 0.999


@pytest.fixture(scope="module")
def backbone(checkpoint_path):
    from mermaid_classifier_amd.backbone import Backbone
    bb = Backbone(str(checkpoint_path), device=0, max_batch=16)
    yield bb
    bb.close()


def test_golden_features_noise(backbone, golden_backbone):
    from oracle import efficientnet_b0_ref as ref
    got = backbone.extract(ref.synthetic_patches(8, seed=42))
    want = golden_backbone["noise8"]
    r, c = rel_l2(got, want), cosine(got, want)
    print("noise rel-L2", r, "cos", c, "maxabs", np.abs(got - want).max())
    assert c.min() >= COS_GATE
    assert r.max() < TOL_NOISE


def test_golden_features_natural(backbone, golden_backbone):
    from oracle import efficientnet_b0_ref as ref
    got = backbone.extract(ref.natural_patches(8, seed=7))
    want = golden_backbone["natural8"]
    r, c = rel_l2(got, want), cosine(got, want)
    print("natural rel-L2", r, "cos", c, "maxabs", np.abs(got - want).max())
    assert c.min() >= COS_GATE
    assert r.max() < TOL_NATURAL


def test_b4_backbone_matches_oracle(synth_sd_b4):
    """BASELINE.json configs[4] (EfficientNet-B4 on 224x224 patches; not in the reference): the generic per-layer
    schedule (stem 48, 32 blocks, squeeze-excite up to 112 units, 2688-channel depthwise, 1792-d features) against the
    oracle's committed fixtures; rows independent of batching.
    Tolerance: B0's gates (1e-3 image-like, 1e-2 white noise, cosine 0.999) -- except where the input is ill-conditioned
    for this random-weight 32-block net: there the oracle's own fp16-storage emulation (every HBM tensor rounded to fp16,
    arithmetic fp32) already misses the gate (1.1e-2 on image-like patch 2, 2e-2 on white noise), and the HIP path must
    stay within twice the emulation's error."""
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    g = np.load(GOLDEN / "backbone_b4_features.npz")
    net = ref.EfficientNetB0Ref(synth_sd_b4, arch="b4")
    bb = Backbone({k: v.numpy() for k, v in synth_sd_b4.items()}, device=0, max_batch=4)
    try:
        assert bb.arch == "b4" and bb.feature_dim == 1792
        nat, noi = ref.natural_patches(4, seed=7), ref.synthetic_patches(4, seed=42)
        outs = []
        for kind, patches, want, tol in (("natural", nat, g["natural4"], TOL_NATURAL), ("noise", noi, g["noise4"], TOL_NOISE)):
            got = bb.extract(patches)
            outs.append(got)
            with torch.no_grad():
                emu = net.extract_features(ref.transformation(patches), emulate_fp16=True).numpy()
            r, e, c = rel_l2(got, want), rel_l2(emu, want), cosine(got, want)
            print(f"b4 {kind} rel-L2 {r} (fp16-storage emulation {e}) cos {c}")
            assert got.shape == (4, 1792)
            # explicit per-row expectations (a second bad row fails): image-like rows 0, 1, 3 meet the gate outright; row 2 -- and
            # the four white-noise rows -- are the inputs on which the oracle's own fp16-storage emulation misses it (1.08e-2;
            # 2e-2), and must stay within twice that emulation's error
            if kind == "natural":
                assert np.all(e[[0, 1, 3]] < tol) and np.all(r[[0, 1, 3]] < tol), (r, e)
                assert r[2] <= 2 * e[2] and np.all(c[[0, 1, 3]] >= COS_GATE), (r, e, c)
            else:
                assert np.all(r <= np.maximum(tol, 2 * e)) and np.all(c >= COS_GATE), (r, e, c)   # the reference's own cosine gate
        both = bb.extract(np.concatenate([nat, noi]))      # 8 > max_batch: internal chunking
        assert np.array_equal(both, np.concatenate(outs))
    finally:
        bb.close()


def test_b4_at_config_batch_size(synth_sd_b4):
    """BASELINE configs[4] at its size (EfficientNet-B4, batch 512 in one pass; fp16 -- fp8 is a study):
    the oracle's golden patches embedded in the batch are reproduced, rows depend only on their own patch (duplicates and a
    permutation give identical bits), the run is deterministic, and 512 patches on a handle built for 512 equal the same
    patches through a handle built for 4."""
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    g = np.load(GOLDEN / "backbone_b4_features.npz")
    base = np.concatenate([ref.natural_patches(4, seed=7), ref.synthetic_patches(4, seed=42), ref.natural_patches(24, seed=17)])
    rng = np.random.default_rng(2)
    idx = rng.integers(0, len(base), size=512)
    idx[:32] = np.arange(32)
    p = base[idx]
    sd = {k: v.numpy() for k, v in synth_sd_b4.items()}
    bb = Backbone(sd, device=0, max_batch=512)
    try:
        f = bb.extract(p)
        assert f.shape == (512, 1792) and np.isfinite(f).all()
        for i in range(32, 512):
            assert np.array_equal(f[i], f[idx[i]])
        perm = rng.permutation(512)
        assert np.array_equal(bb.extract(p[perm]), f[perm])
        assert np.array_equal(bb.extract(p), f)
        r = rel_l2(f[:4], g["natural4"])
        print("b4 @512 natural rel-L2", r, "noise cos", cosine(f[4:8], g["noise4"]))
        # per-row, as in test_b4_backbone_matches_oracle: rows 0, 1, 3 meet the gate; row 2 (fp16-storage emulation 1.08e-2,
        # DESIGN.md "Numerics") stays within twice the emulated error
        assert np.all(r[[0, 1, 3]] < TOL_NATURAL) and r[2] <= 2 * 1.08e-2, r
        assert cosine(f[4:8], g["noise4"]).min() >= COS_GATE
    finally:
        bb.close()
    small = Backbone(sd, device=0, max_batch=4)
    try:
        assert np.array_equal(small.extract(p[:12]), f[:12])
    finally:
        small.close()


def test_b4_fp8_project_convs_on_e4m3_operands(synth_sd_b4):
    """BASELINE.json configs[4] as stated: EfficientNet-B4 with fp8 (OCP e4m3) weights / activations on the CDNA4 fp8 MFMA
    (precision="fp8": the squeeze-excite gated project convs of the 7x7 stage, per-output-channel weight scales, per-pixel
    activation scales; pw_gemm_fp8_kernel).  Not in the reference.  Accuracy is reported and bounded by what the operand format
    itself costs: the fp32 oracle re-run with exactly these operands quantised (tests/study_fp8.py) gives rel-L2 2.7e-2 .. 5.8e-2
    and cosine >= 0.9983 on the image-like patches, 6.3e-2 .. 8.1e-2 / >= 0.9969 on white noise.  Gates: cosine >= 0.997
    (image-like) / >= 0.995 (noise) against the fp32 oracle; rel-L2 within 1.3 x the emulation's + the fp16 path's own error.
    Rows stay independent of batching."""
    import torch
    import study_fp8
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    g = np.load(GOLDEN / "backbone_b4_features.npz")
    net = ref.EfficientNetB0Ref(synth_sd_b4, arch="b4")
    sd = {k: v.numpy() for k, v in synth_sd_b4.items()}
    bb = Backbone(sd, device=0, max_batch=4, precision="fp8")
    bh = Backbone(sd, device=0, max_batch=4)
    try:
        outs = []
        for kind, patches, want, cos_gate in (("natural", ref.natural_patches(4, seed=7), g["natural4"], 0.997),
                                              ("noise", ref.synthetic_patches(4, seed=42), g["noise4"], 0.995)):
            got, half = bb.extract(patches), bh.extract(patches)
            outs.append(got)
            with torch.no_grad():
                emu = study_fp8.forward(net, ref.transformation(patches), "pixel", lambda k, i, h: k == "project" and h <= 7).numpy()
            r, e, rh, c = rel_l2(got, want), rel_l2(emu, want), rel_l2(half, want), cosine(got, want)
            print(f"b4 fp8 {kind}: rel-L2 {r} (operand-format emulation {e}, fp16 path {rh}) cos {c}")
            assert np.all(c >= cos_gate), (kind, c)
            assert np.all(r <= 1.3 * e + rh), (kind, r, e, rh)
            assert not np.array_equal(got, half)          # the fp8 path really ran
        both = bb.extract(np.concatenate([ref.natural_patches(4, seed=7), ref.synthetic_patches(4, seed=42)]))
        assert np.array_equal(both, np.concatenate(outs))
        # ragged row counts (1, 3 patches: 49 / 147 GEMM rows, partial 64-row workgroups) and an empty call
        nat = ref.natural_patches(4, seed=7)
        assert np.array_equal(bb.extract(nat[:1]), outs[0][:1]) and np.array_equal(bb.extract(nat[1:4]), outs[0][1:4])
        assert bb.extract(nat[:0]).shape == (0, 1792)
    finally:
        bb.close()
        bh.close()
    with pytest.raises(ValueError):
        Backbone({k: v for k, v in sd.items()}, device=0, max_batch=4, precision="int4")


def test_batching_is_bitwise_invariant(backbone):
    """Rows are independent: any split of the same patches gives identical bits (ragged + max-batch)."""
    from oracle import efficientnet_b0_ref as ref
    p = ref.natural_patches(19, seed=3)          # 19 > max_batch=16 -> internal chunking, ragged tail
    whole = backbone.extract(p)
    parts = np.concatenate([backbone.extract(p[:1]), backbone.extract(p[1:8]), backbone.extract(p[8:])])
    assert np.array_equal(whole, parts)
    assert backbone.extract(p[:0]).shape == (0, 1280)
    again = backbone.extract(p)
    assert np.array_equal(whole, again)           # deterministic run to run


def test_device_resident_path_matches_host_path(backbone):
    import torch
    from oracle import efficientnet_b0_ref as ref
    p = ref.synthetic_patches(5, seed=11)
    host = backbone.extract(p)
    dev = backbone.extract(torch.from_numpy(p).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(host, dev.cpu().numpy())


def test_pipelined_multi_chunk_call_is_bitwise_the_chunks(backbone):
    """A device-resident call with n > max_batch is ONE pass pipelined over max_batch-sized chunks (each lane walks its
    sub-batches without waiting for the others; one fork/join): bits equal those of separate per-chunk calls, plain
    launches and graph replay alike (the third identical call replays the captured graph)."""
    import torch
    from oracle import efficientnet_b0_ref as ref
    p = torch.from_numpy(ref.natural_patches(53, seed=21)).cuda()     # max_batch 16: chunks 16,16,16,5 (ragged, odd)
    out = torch.empty((53, 1280), dtype=torch.float32, device="cuda")
    runs = []
    for _ in range(4):
        out.zero_()
        backbone.extract(p, out=out)
        torch.cuda.synchronize()
        runs.append(out.cpu().numpy().copy())
    parts = np.concatenate([backbone.extract(p[i:i + 16]).cpu().numpy() for i in range(0, 53, 16)])
    for r in runs:
        assert np.array_equal(r, parts)


def test_extractor_contract_and_image_path(checkpoint_path, oracle_net, golden_backbone):
    """The reference's two call shapes: patches_to_features(list[PIL]) and extractor(image, rowcols)."""
    from PIL import Image
    from mermaid_classifier_amd import build_extractor_class, verify_device_numerics
    from mermaid_classifier_amd.spacer_shim import DataLocation
    from oracle import efficientnet_b0_ref as ref
    cls = build_extractor_class()
    ex = cls(data_locations={"weights": DataLocation("filesystem", str(checkpoint_path))}, device="cuda", batch_size=10)
    patches = [Image.fromarray(a) for a in ref.natural_patches(3, seed=7)]
    feats, loaded_remote = ex.patches_to_features(patches)
    assert loaded_remote is False and isinstance(feats, list) and isinstance(feats[0][0], float)
    assert np.asarray(feats).shape == (3, 1280)
    assert rel_l2(np.asarray(feats), golden_backbone["natural8"][:3]).max() < TOL_NATURAL
    # image + rowcols (config 1 geometry scaled down; includes corner points that exercise reflect padding)
    rng = np.random.default_rng(42)
    image = rng.integers(0, 255, (696, 928, 3), dtype=np.uint8)
    rowcols = [tuple(int(v) for v in rc) for rc in golden_backbone["image_rowcols"]]
    features, msg = ex(Image.fromarray(image), rowcols)
    got = np.vstack([features.get_array(rc) for rc in rowcols])
    want = golden_backbone["image_features"]
    assert features.npoints == len(rowcols) and features.feature_dim == 1280 and msg.runtime > 0
    assert cosine(got, want).min() >= COS_GATE and rel_l2(got, want).max() < TOL_NOISE
    # the reference's own gate, with the oracle as the CPU side (reference signature; keyword-only substitute for pyspacer)
    loc = DataLocation("filesystem", str(checkpoint_path))
    assert verify_device_numerics(ex, loc, 10, "cuda",
                                  cpu_features_fn=lambda ps: ref.patches_to_features(oracle_net, np.stack(ps))) is None
    with pytest.raises(ValueError):
        ex(Image.fromarray(image), [(700, 5)])


def test_numerics_gate_against_stock_cpu_path(checkpoint_path, caplog):
    """scripts/build_feature_bucket.py:854-861 unmodified: the class factory's product on the GPU, then
    ``verify_device_numerics(extractor, weights_loc, batch_size, device)`` -- whose CPU side is ``cls(device="cpu")`` from
    the same factory, i.e. pyspacer's own torch-CPU path (a stand-in pyspacer backed by the oracle is installed)."""
    import logging
    import fake_spacer
    from mermaid_classifier_amd import build_extractor_class, verify_device_numerics
    from mermaid_classifier_amd.spacer_shim import DataLocation
    loc = DataLocation("filesystem", str(checkpoint_path))
    with fake_spacer.installed() as Base:
        cls = build_extractor_class()
        assert issubclass(cls, Base)
        extractor = cls(data_locations={"weights": loc}, device="cuda", batch_size=10)
        with caplog.at_level(logging.INFO, logger="mermaid_classifier_amd.extractor"):
            assert verify_device_numerics(extractor, loc, 10, "cuda") is None
        print(caplog.text)
        assert "min_cos=0.9999" in caplog.text
        # pyspacer's __call__ (CPU crop, PIL patches) feeding the HIP patches_to_features
        rng = np.random.default_rng(5)
        image = rng.integers(0, 255, (500, 640, 3), dtype=np.uint8)
        feats, _ = extractor(image, [(10, 20), (250, 320)])
        assert feats.npoints == 2 and len(feats.point_features[0].data) == 1280
        extractor._cached_net.close()


def test_crop_kernel_matches_oracle_bitwise():
    from mermaid_classifier_amd.backbone import crop_patches_device
    from oracle import pyspacer_ref
    rng = np.random.default_rng(5)
    image = rng.integers(0, 256, (300, 411, 3), dtype=np.uint8)
    rowcols = [(0, 0), (299, 410), (0, 410), (299, 0), (150, 200), (1, 409), (111, 112), (113, 300)]
    got = crop_patches_device(image, rowcols).cpu().numpy()
    want = pyspacer_ref.crop_patches(image, rowcols, 224)
    assert np.array_equal(got, want)


def test_sparse_points_are_cut_on_the_host_bitwise():
    """Few points on a big host image (the reference's data: 10-25 points per 27 MP image) are cut on the host into a pinned
    ring slot and only the patches are uploaded; same bits as the oracle, borders included, ring reuse included."""
    import torch
    from mermaid_classifier_amd.backbone import crop_patches_device
    from oracle import pyspacer_ref
    rng = np.random.default_rng(6)
    image = rng.integers(0, 256, (1300, 1500, 3), dtype=np.uint8)          # 5.85 MB: up to 6 points take the host path
    sets = [[(0, 0), (1299, 1499), (0, 1499), (1299, 0), (650, 700)], [(111, 112), (113, 1387), (1188, 5)], [(640, 1388)],
            [(5, 5), (1294, 1494)], [(700, 111), (700, 112), (112, 700), (111, 700)], [(1187, 1387)]]
    outs = [crop_patches_device(image, rc) for rc in sets]                  # 6 calls: wraps the 4-slot ring
    torch.cuda.synchronize()
    for rc, got in zip(sets, outs):
        assert np.array_equal(got.cpu().numpy(), pyspacer_ref.crop_patches(image, rc, 224))
    dense = [(int(r), int(c)) for r, c in zip(rng.integers(0, 1300, 40), rng.integers(0, 1500, 40))]   # crop_kernel path
    assert np.array_equal(crop_patches_device(image, dense).cpu().numpy(), pyspacer_ref.crop_patches(image, dense, 224))


@pytest.mark.parametrize("name", ["head_fixture", "head108"])
def test_head_matches_reference_predictor(name):
    """HIP head vs outputs of the reference's own CalibratedHead/Predictor (golden, generated by
    importing the reference): max|dp| <= 1e-6 (the reference's export gate, inference/export.py:31)
    and identical argmax on every row."""
    from mermaid_classifier_amd import load_predictor
    io = np.load(GOLDEN / f"{name}_io.npz")
    pred = load_predictor(GOLDEN / name / "model.pt", GOLDEN / name / "model.json")
    got = pred.predict_proba(io["X"])
    want = io["proba_predictor_f64"]
    assert got.dtype == np.float64 and got.shape == want.shape
    d = np.abs(got - want).max()
    print(name, "max|dp|", d)
    assert d <= (1e-6 if name == "head_fixture" else 2e-5)
    assert np.array_equal(got.argmax(1), want.argmax(1))
    assert np.abs(got.sum(1) - 1).max() < 1e-5
    labels = pred.predict(io["X"])
    assert labels == [pred.classes[i] for i in want.argmax(1)]
    with pytest.raises(ValueError):
        pred.predict_proba(np.zeros((2, pred.input_dim + 1), np.float32))
    assert pred.predict_proba(np.zeros((0, pred.input_dim), np.float32)).shape == (0, len(pred.classes))


def test_extract_then_classify_labels_match_oracle_chain(backbone, oracle_net):
    """Config 3 in miniature: HIP features -> HIP head labels == oracle features -> reference-restated head labels."""
    from mermaid_classifier_amd import load_predictor
    from oracle import efficientnet_b0_ref as ref, head_ref
    from mermaid_classifier_amd.inference import params_from_torchscript
    import torch
    pred = load_predictor(GOLDEN / "head108" / "model.pt", GOLDEN / "head108" / "model.json")
    p = ref.natural_patches(12, seed=21)
    f_hip = backbone.extract(p)
    f_ref = ref.patches_to_features(oracle_net, p)
    prm = params_from_torchscript(torch.jit.load(str(GOLDEN / "head108" / "model.pt")))
    want = head_ref.predict_proba(f_ref, prm.weights, prm.biases, prm.a, prm.b, 1280)
    got = pred.predict_proba(f_hip)
    # every row is accounted for (conftest.check_labels): identical wherever the reference's top-2 margin exceeds twice the
    # measured probability perturbation; flips inside that band are counted, must land on the runner-up, and are bounded
    check_labels(got, want, dp_bound=2e-4, max_flips=1, what="12 image-like patches, head108")
    # same features in -> identical labels, always
    assert np.array_equal(pred.predict_proba(f_ref).argmax(1), want.argmax(1))


def test_cross_image_batching_matches_per_image_oracle(backbone, oracle_net):
    """BASELINE config 3 in miniature: several images x several points through the cross-image batcher
    (GPU crop + large passes) == the oracle's per-image crop + forward, image by image, point order kept;
    ragged point counts, an image without points, and a buffer smaller than one image's points."""
    from mermaid_classifier_amd.pipeline import BatchedExtractor
    from oracle import pyspacer_ref
    rng = np.random.default_rng(11)
    images = [rng.integers(0, 255, (300 + 17 * i, 420 - 11 * i, 3), dtype=np.uint8) for i in range(4)]
    rowcols = [[(0, 0), (150, 200), (299, 419)], [], [(10, 20), (300, 5), (7, 390), (160, 160), (333, 397)],
               [(int(r), int(c)) for r, c in zip(rng.integers(0, 351, 9), rng.integers(0, 387, 9))]]
    ex = BatchedExtractor(backbone, batch_patches=6)       # forces flushes inside an image
    got = ex.extract_images(images, rowcols)
    assert [g.shape for g in got] == [(3, 1280), (0, 1280), (5, 1280), (9, 1280)]
    for im, rc, g in zip(images, rowcols, got):
        if rc:
            want = pyspacer_ref.extract(oracle_net, im, rc)
            assert cosine(g, want).min() >= COS_GATE and rel_l2(g, want).max() < TOL_NOISE
    feats = ex.extract_image_features(images[:1], rowcols[:1])[0]
    assert feats.npoints == 3 and np.allclose(feats.get_array((150, 200)), got[0][1])
    with pytest.raises(ValueError):
        ex.extract_images(images[:1], [[(400, 1)]])


def test_full_batch_size_independent_properties(checkpoint_path, golden_backbone):
    """BASELINE configs[1] size (256 patches per pass, both lanes, every CU busy): properties that do not need
    the oracle at this size -- rows depend only on their own patch (duplicates and permutations are bitwise
    consistent), run-to-run determinism, and the golden rows are reproduced inside the big batch."""
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    base = np.concatenate([ref.natural_patches(8, seed=7), ref.synthetic_patches(8, seed=42), ref.natural_patches(16, seed=13)])
    rng = np.random.default_rng(1)
    idx = rng.integers(0, len(base), size=256)
    idx[:32] = np.arange(32)
    p = base[idx]
    bb = Backbone(str(checkpoint_path), device=0, max_batch=256)
    f = bb.extract(p)
    assert np.isfinite(f).all() and f.shape == (256, 1280)
    for i in range(32, 256):                                  # duplicates of a patch give the duplicate's exact bits
        assert np.array_equal(f[i], f[idx[i]])
    perm = rng.permutation(256)
    assert np.array_equal(bb.extract(p[perm]), f[perm])       # permutation equivariance, bitwise
    assert np.array_equal(bb.extract(p), f)                   # deterministic
    assert rel_l2(f[:8], golden_backbone["natural8"]).max() < TOL_NATURAL
    assert cosine(f[8:16], golden_backbone["noise8"]).min() >= COS_GATE
    # 300 > max_batch: internal chunking + ragged lanes give the same rows
    q = np.concatenate([p, p[:44]])
    g = bb.extract(q)
    assert np.array_equal(g[:256], f) and np.array_equal(g[256:], f[:44])
    bb.close()


@pytest.mark.parametrize("knob", ["MMC_FUSE_B0", "MMC_SE_SMALL", "MMC_PROJSE", "MMC_TAIL_B11", "MMC_TAIL_FULL", "MMC_TAIL", "MMC_MB_DOT2", "MMC_MID14", "MMC_MID14=2", "MMC_MID14_B11=1", "MMC_MID14M=1", "MMC_MID14M=0", "MMC_MBT4=0", "MMC_TAIL_DW4=1", "MMC_MB1", "MMC_MBT", "MMC_MBT2", "MMC_THIN_PROJ", "MMC_B1_PLANAR", "MMC_GRAPH",
                                  "MMC_LANES"])
def test_every_schedule_variant_meets_the_same_gates(checkpoint_path, golden_backbone, knob, monkeypatch):
    """Each fusion has an environment switch (the separate kernels stay in the library as the reference
    schedule).  With any one of them off -- or a single lane -- the features must still pass the golden gates, and
    the default schedule must agree with the variant to within fp16 rounding of the intermediates."""
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    p = ref.natural_patches(8, seed=7)
    bb = Backbone(str(checkpoint_path), device=0, max_batch=8)
    base = bb.extract(p)
    bb.close()
    if "=" in knob:
        monkeypatch.setenv(*knob.split("="))
    else:
        monkeypatch.setenv(knob, "1" if knob == "MMC_LANES" else "0")
    bb = Backbone(str(checkpoint_path), device=0, max_batch=8)
    try:
        got = bb.extract(p)
    finally:
        bb.close()
    want = golden_backbone["natural8"]
    assert rel_l2(got, want).max() < TOL_NATURAL and cosine(got, want).min() >= COS_GATE
    assert rel_l2(got, base).max() < TOL_NATURAL
    if knob == "MMC_LANES":
        assert np.array_equal(got, base)          # lanes only split the batch: bitwise identical


@pytest.mark.parametrize("knob", ["MMC_THIN_PROJ", "MMC_PROJSE", "MMC_FUSE", "MMC_B4_CC14=48", "MMC_LANES"])
def test_b4_schedule_variants(synth_sd_b4, knob, monkeypatch):
    """B4's schedule switches: lanes only split the batch (bitwise equal); thin_proj (gate folded into the weight fragments) /
    proj_patch / the fused expand+depthwise / the chunk width change rounding points only."""
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    sd = {k: v.numpy() for k, v in synth_sd_b4.items()}
    p = ref.natural_patches(4, seed=7)[[0, 1, 3]]          # the well-conditioned inputs (see test_b4_backbone_matches_oracle)
    bb = Backbone(sd, device=0, max_batch=4)
    base = bb.extract(p)
    bb.close()
    if "=" in knob:
        monkeypatch.setenv(*knob.split("="))
    else:
        monkeypatch.setenv(knob, "1" if knob == "MMC_LANES" else "0")
    bb = Backbone(sd, device=0, max_batch=4)
    try:
        got = bb.extract(p)
    finally:
        bb.close()
    if knob == "MMC_LANES":
        assert np.array_equal(got, base)
    else:
        assert rel_l2(got, base).max() < TOL_NATURAL
    want = np.load(GOLDEN / "backbone_b4_features.npz")["natural4"][[0, 1, 3]]
    assert rel_l2(got, want).max() < TOL_NATURAL and cosine(got, want).min() >= COS_GATE


def test_graph_replay_is_bitwise_identical_to_plain_launches(checkpoint_path, monkeypatch):
    """A pass over buffers that repeat is captured into a HIP graph on the third call and replayed afterwards
    (mmc_api.cpp run_pass); replay, plain launches (MMC_GRAPH=0) and a call with fresh buffers must agree bitwise."""
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    p = torch.from_numpy(ref.natural_patches(12, seed=5)).cuda()
    out = torch.empty((12, 1280), dtype=torch.float32, device="cuda")
    bb = Backbone(str(checkpoint_path), device=0, max_batch=12)
    runs = []
    for _ in range(6):                       # same buffers every time: plain, plain, capture + launch, replay ...
        out.zero_()
        bb.extract(p, out=out)
        runs.append(out.cpu().numpy().copy())
    fresh = bb.extract(p.clone()).cpu().numpy()
    bb.close()
    monkeypatch.setenv("MMC_GRAPH", "0")
    bb = Backbone(str(checkpoint_path), device=0, max_batch=12)
    plain = bb.extract(p).cpu().numpy()
    bb.close()
    for r in runs:
        assert np.array_equal(r, plain)
    assert np.array_equal(fresh, plain)


def test_graph_cache_eviction_and_stream_switches_keep_results(checkpoint_path):
    """include/mmc.h: the 32 most recently used (patches, features, n) combinations keep their HIP graph, older ones are
    evicted and fall back to plain launches until they repeat; a call on a different stream than the previous one first waits
    for that call's work (shared workspace).  40 distinct input buffers, each seen three times (capture on the third), on two
    alternating streams: every result equals the plain-launch result of the same patches."""
    import torch
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    base = torch.from_numpy(ref.natural_patches(6, seed=5)).cuda()
    bb = Backbone(str(checkpoint_path), device=0, max_batch=8)
    try:
        want = bb.extract(base).clone()
        bufs = [base.roll(i % 6, 0).clone() for i in range(40)]
        outs = [torch.empty((6, 1280), dtype=torch.float32, device="cuda") for _ in range(40)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        for rep in range(3):
            for i, (p, o) in enumerate(zip(bufs, outs)):
                with torch.cuda.stream(streams[(i + rep) & 1]):
                    bb.extract(p, out=o)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, want.roll(i % 6, 0)), i
        # Cost, not only results: a caller that cycles through more combinations than the cache holds must not pay a capture
        # per call.  A combination whose graph was evicted is never captured again, and after one turnover of the cache
        # (32 evictions) nothing new is captured: the counters stop moving however many more rounds are run.
        st0 = bb.graph_stats()
        assert st0["cached"] <= 32 and st0["captures"] <= 32 + 32
        for rep in range(3):
            for i, (p, o) in enumerate(zip(bufs, outs)):
                bb.extract(p, out=o)
        torch.cuda.synchronize()
        st1 = bb.graph_stats()
        assert st1["captures"] <= 32 + 32 and st1["evictions"] <= 32, (st0, st1)
        for i, o in enumerate(outs):
            assert torch.equal(o, want.roll(i % 6, 0)), i
    finally:
        bb.close()


def test_phase_clock_build_gives_the_same_features(checkpoint_path, monkeypatch):
    """MMC_TAIL_CLK=1 (the in-kernel phase clocks behind profiles/r02_tail_phases.txt) must not change a bit of the result, and
    the clock buffer is filled for every workgroup."""
    from mermaid_classifier_amd.backbone import Backbone
    from oracle import efficientnet_b0_ref as ref
    p = ref.natural_patches(6, seed=9)
    bb = Backbone(str(checkpoint_path), device=0, max_batch=8)
    plain = bb.extract(p)
    bb.close()
    monkeypatch.setenv("MMC_TAIL_CLK", "1")
    bc = Backbone(str(checkpoint_path), device=0, max_batch=8)
    try:
        assert np.array_equal(bc.extract(p), plain)
        clk = bc.read_activation("tail.clk", 8 * 64).reshape(8, 8, 8)     # rows: lane 0's patches, then lane 1's from row 4
        ran = clk[:, 6, 0] > 1e4                                           # whole-kernel cycles
        assert ran.sum() == 6
        assert (clk[ran][:, 1:5, :6] > 0).all()                            # every phase of blocks 12..15
    finally:
        bc.close()


def test_plain_c_host_gets_the_same_features(tmp_path, checkpoint_path):
    """examples/c_host.c -- a C99 program with nothing but include/mmc.h and libmermaid_mi355.so (host buffers, no torch, no HIP calls
    of its own) -- built with gcc and run on the packed weights: its features are bit for bit those of the Python path."""
    import shutil
    import subprocess
    from mermaid_classifier_amd import weights as W
    from mermaid_classifier_amd.backbone import Backbone
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    root = Path(__file__).resolve().parent.parent
    libdir = root / "mermaid_classifier_amd"
    exe = tmp_path / "c_host"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I", str(root / "include"), str(root / "examples" / "c_host.c"),
                        "-o", str(exe), "-L", str(libdir), "-lmermaid_mi355", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with open(checkpoint_path, "rb") as f:
        blob = W.pack_from_stream(f, W.get_arch(None))
    (tmp_path / "w.mmcw").write_bytes(blob)
    patches = np.random.default_rng(42).integers(0, 255, (5, 224, 224, 3), dtype=np.uint8)
    (tmp_path / "p.u8").write_bytes(patches.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "w.mmcw"), str(tmp_path / "p.u8"), "5", str(tmp_path / "f.f32")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "5 patches -> (5, 1280) features" in r.stdout
    got = np.fromfile(tmp_path / "f.f32", dtype=np.float32).reshape(5, 1280)
    bb = Backbone(str(checkpoint_path), device=0, max_batch=64)
    try:
        want = bb.extract(patches)
    finally:
        bb.close()
    np.testing.assert_array_equal(got, np.asarray(want))
