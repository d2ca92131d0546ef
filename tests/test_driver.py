"""Per-source driver: bookkeeping semantics of the reference loop
(scripts/build_feature_bucket.py:691-823) over cross-image GPU batches."""
import csv
import io
import json

import numpy as np
import pytest

from mermaid_classifier_amd import driver
from mermaid_classifier_amd.spacer_shim import DataLocation, ImageFeatures, PointFeatures


class _FakeBatched:
    """Duck-typed BatchedExtractor for the host-logic test: feature = [row, col, mean pixel]."""
    feature_dim = 3

    def __init__(self, explode_on=None):
        self.calls = []
        self.explode_on = explode_on

    def extract_image_features(self, images, rowcols_per_image):
        self.calls.append(len(images))
        out = []
        for im, rc in zip(images, rowcols_per_image):
            if self.explode_on is not None and int(im[0, 0, 0]) == self.explode_on:
                raise RuntimeError("device lost")
            pfs = [PointFeatures(r, c, [float(r), float(c), float(im.mean())]) for r, c in rc]
            out.append(ImageFeatures(pfs, True, 3, len(pfs)))
        return out


def _img(v, h=300, w=320):
    return np.full((h, w, 3), v, np.uint8)


def test_process_source_bookkeeping(tmp_path):
    images = {
        "0003": [(10, 10), (200, 300)],
        "0001": [(5, 5)],
        "0002": [],                       # no points -> skipped(no_rowcols)
        "0004": [(299, 319)],             # already extracted -> skipped(exists)
        "0005": [(400, 10)],              # point outside the image -> failed(ValueError), isolated
        "0006": [(1, 1)],                 # load error -> failed(OSError)
        "0007": [(7, 7)],
    }
    pixels = {"0001": 1, "0003": 3, "0005": 5, "0007": 7}

    def load(image_id):
        if image_id == "0006":
            raise OSError("truncated jpeg")
        return _img(pixels[image_id])

    root = str(tmp_path)
    store = driver.fs_store(root, "17")
    store("0004", ImageFeatures([PointFeatures(299, 319, [0.0, 0.0, 0.0])], True, 3, 1))
    assert driver.fs_existing(root, "17") == {"0004"}
    prog, errf = io.StringIO(), io.StringIO()
    ew = csv.writer(errf)
    ex = _FakeBatched()
    c = driver.process_source(source_id="17", images=images, load_image=load, store_features=store, extractor=ex,
                              existing=driver.fs_existing(root, "17"), progress_writer=prog, error_writer=ew,
                              prefetch=2, group_images=2)
    assert (c.images_ok, c.images_skipped, c.images_failed, c.sources_done) == (3, 2, 2, 1)
    assert ex.calls == [2, 1]                                    # grouped passes, not one per image
    recs = [json.loads(l) for l in prog.getvalue().splitlines()]
    by = {r["image_id"]: r for r in recs}
    assert by["0002"]["outcome"] == "skipped" and by["0002"]["reason"] == "no_rowcols"
    assert by["0004"]["outcome"] == "skipped" and by["0004"]["reason"] == "exists"
    assert by["0005"]["outcome"] == "failed" and by["0005"]["error_type"] == "ValueError"
    assert by["0006"]["outcome"] == "failed" and by["0006"]["error_type"] == "OSError"
    assert all(set(r) >= {"ts", "source_id", "image_id", "outcome"} and r["source_id"] == "17" for r in recs)
    rows = list(csv.reader(io.StringIO(errf.getvalue())))
    assert [r[2] for r in rows] == ["0005", "0006"] and rows[0][3] == "ValueError" and "outside" in rows[0][4]
    assert driver.fs_existing(root, "17") == {"0001", "0003", "0004", "0007"}
    f3 = ImageFeatures.load(DataLocation("filesystem", str(tmp_path / driver.feature_key("17", "0003"))))
    np.testing.assert_allclose(f3.get_array((200, 300)), [200.0, 300.0, 3.0])
    # resume: a second run touches nothing
    ex2 = _FakeBatched()
    c2 = driver.process_source(source_id="17", images={k: v for k, v in images.items() if k not in ("0005", "0006")},
                               load_image=load, store_features=store, extractor=ex2, existing=driver.fs_existing(root, "17"))
    assert (c2.images_ok, c2.images_skipped, c2.images_failed) == (0, 5, 0) and ex2.calls == []


def test_group_failure_and_dry_run(tmp_path):
    images = {"a": [(1, 1)], "b": [(2, 2)], "c": [(3, 3)]}
    load = lambda i: _img({"a": 1, "b": 2, "c": 3}[i])          # noqa: E731
    stored = {}
    ex = _FakeBatched(explode_on=2)
    c = driver.process_source(source_id="1", images=images, load_image=load, store_features=stored.__setitem__,
                              extractor=ex, group_images=2)
    # group (a, b) dies on the device -> both failed; c is unaffected
    assert (c.images_ok, c.images_failed) == (1, 2) and set(stored) == {"c"}
    c = driver.process_source(source_id="1", images=images, load_image=load, store_features=stored.__setitem__,
                              extractor=_FakeBatched(), dry_run=True)
    assert c.images_ok == 3 and set(stored) == {"c"}
    assert driver.process_source(source_id="2", images={}, load_image=load, store_features=stored.__setitem__,
                                 extractor=_FakeBatched()).sources_skipped == 1


def test_open_logs_header(tmp_path):
    pf, ew, close = driver.open_logs(str(tmp_path / "p.jsonl"), str(tmp_path / "e.csv"))
    driver.record_failure(ew, "1", "2", "ValueError", "x")
    driver.record_progress(pf, "1", "2", "failed", error_type="ValueError")
    close()
    rows = list(csv.reader(open(tmp_path / "e.csv")))
    assert rows[0] == ["ts", "source_id", "image_id", "error_type", "error_msg"] and rows[1][1:] == ["1", "2", "ValueError", "x"]


@pytest.mark.gpu
def test_driver_on_gpu_matches_direct_extraction(tmp_path, synth_sd):
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.pipeline import BatchedExtractor
    from oracle import pyspacer_ref
    rng = np.random.default_rng(5)
    imgs = {f"{i:03d}": rng.integers(0, 255, (400 + 16 * i, 500, 3), dtype=np.uint8) for i in range(5)}
    pts = {k: [(int(rng.integers(0, v.shape[0])), int(rng.integers(0, v.shape[1]))) for _ in range(7)] for k, v in imgs.items()}
    pts["003"].append((10_000, 3))                                # this image fails validation, the others are extracted
    bb = Backbone(synth_sd, device=0, max_batch=64)
    try:
        c = driver.process_source(source_id="9", images=pts, load_image=imgs.__getitem__, store_features=driver.fs_store(str(tmp_path), "9"),
                                  extractor=BatchedExtractor(bb, batch_patches=64), group_images=3)
        assert (c.images_ok, c.images_failed) == (4, 1)
        for k in ("000", "004"):
            got = ImageFeatures.load(DataLocation("filesystem", str(tmp_path / driver.feature_key("9", k))))
            patches = pyspacer_ref.crop_patches(imgs[k], pts[k])
            want = bb.extract(np.stack(patches))
            for (r, cc), row in zip(pts[k], want):
                np.testing.assert_array_equal(got.get_array((r, cc)), row.astype(np.float32))
    finally:
        bb.close()


@pytest.mark.gpu
def test_batched_extractor_sparse_points_many_flushes_bitwise(synth_sd):
    """Config-3 geometry in the regime the reference's data lives in: big decoded images, few points each -> the patches are
    cut on the host into pinned slots and uploaded on the copy stream, two patch buffers alternate, results are collected per
    pass through events.  With a 40-patch buffer the 9 images below take seven flushes; an image may span two flushes; one
    image has no points; one has more points than the buffer.  Every feature row must equal the direct extraction of the
    oracle's crop, bit for bit, in rowcols order."""
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.pipeline import BatchedExtractor
    from oracle import pyspacer_ref
    rng = np.random.default_rng(8)
    base = [rng.integers(0, 255, (2900, 3000, 3), dtype=np.uint8) for _ in range(2)]     # 26 MB: <= 28 points take the host cut
    images = [base[i % 2] for i in range(9)]
    counts = [25, 25, 0, 13, 25, 47, 1, 25, 25]                                          # 47 > 40: spans flushes
    rcs = [[(int(rng.integers(0, 2900)), int(rng.integers(0, 3000))) for _ in range(n)] for n in counts]
    rcs[0][:4] = [(0, 0), (2899, 2999), (0, 2999), (2899, 0)]                            # corners: reflect padding
    bb = Backbone(synth_sd, device=0, max_batch=32)
    try:
        bx = BatchedExtractor(bb, batch_patches=40)
        got = bx.extract_images(images, rcs)
        again = bx.extract_images(images, rcs)                                           # buffers / events are reused across calls
        assert [g.shape for g in got] == [(n, 1280) for n in counts]
        for im, rc, g, g2 in zip(images, rcs, got, again):
            assert np.array_equal(g, g2)
            if rc:
                want = bb.extract(pyspacer_ref.crop_patches(im, rc))
                assert np.array_equal(g, want)
    finally:
        bb.close()


@pytest.mark.gpu
def test_config3_chain_driver_to_npy_to_labels(tmp_path, synth_sd, oracle_net):
    """BASELINE config 3 end to end, in miniature: images x 25 points (the 5x5 grid geometry of the reference's
    docs/pyspacer/0032dba6_points.csv, scaled) -> driver (GPU crop, cross-image batches) -> .featurevector files ->
    extract_reference_features stacking (file order, then point order) -> (N, 1280) .npy -> calibrated head.
    Checked against the oracle chain: per-image crop + fp32 forward, stacked the same way, reference-restated head."""
    import torch
    from conftest import GOLDEN, cosine, rel_l2
    from mermaid_classifier_amd import load_predictor
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.extract_reference_features import main as stack_main
    from mermaid_classifier_amd.inference import params_from_torchscript
    from mermaid_classifier_amd.pipeline import BatchedExtractor
    from oracle import head_ref, pyspacer_ref
    rng = np.random.default_rng(3)
    H, W = 609, 696                                        # 4872 x 5568 / 8
    grid = [(int(H * (2 * i + 1) / 10), int(W * (2 * j + 1) / 10)) for i in range(5) for j in range(5)]
    imgs = {}
    from scipy.ndimage import zoom
    for i in range(3):                                     # smooth (image-like) content: bilinearly upsampled noise
        base = rng.integers(0, 255, (H // 16 + 2, W // 16 + 2, 3)).astype(np.float32)
        imgs[f"{i:04d}"] = np.clip(zoom(base, (16, 16, 1), order=1)[:H, :W], 0, 255).astype(np.uint8)
    pts = {k: list(grid) for k in imgs}
    bb = Backbone(synth_sd, device=0, max_batch=32)
    try:
        c = driver.process_source(source_id="3", images=pts, load_image=imgs.__getitem__, store_features=driver.fs_store(str(tmp_path), "3"),
                                  extractor=BatchedExtractor(bb, batch_patches=32))
    finally:
        bb.close()
    assert (c.images_ok, c.images_failed) == (3, 0)
    files = [str(tmp_path / driver.feature_key("3", k)) for k in sorted(imgs)]
    out = tmp_path / "reference_features.npy"
    stack_main(["--out", str(out)] + files)
    got = np.load(out)
    assert got.shape == (75, 1280) and got.dtype == np.float32
    want = pyspacer_ref.stack_reference_features([pyspacer_ref.extract(oracle_net, imgs[k], pts[k]) for k in sorted(imgs)])
    assert cosine(got, want).min() >= 0.999 and rel_l2(got, want).max() < 1e-2
    pred = load_predictor(GOLDEN / "head108" / "model.pt", GOLDEN / "head108" / "model.json")
    prm = params_from_torchscript(torch.jit.load(str(GOLDEN / "head108" / "model.pt")))
    p_ref = head_ref.predict_proba(want, prm.weights, prm.biases, prm.a, prm.b, 1280)
    p_got = pred.predict_proba(got)
    from conftest import check_labels
    check_labels(p_got, p_ref, dp_bound=2e-4, max_flips=1, what="75 patches of the config-3 chain, head108")


@pytest.mark.gpu
def test_config3_sized_run_through_the_driver(synth_sd, oracle_net):
    """BASELINE configs[2] at its stated size: 10 000 images x 25 points (the reference's 5x5 grid, docs/pyspacer/0032dba6_points.csv
    geometry scaled by 8; MMC_TEST_NIMG overrides the image count) through driver.process_source -> cross-image batches of
    1 024 patches -> 250 000 feature rows,
    stacked in extract_reference_features order (sorted image id, then point order).  The oracle cannot follow at this
    size, so beyond two oracle-checked images the run is held to size-independent properties: every image accounted
    for, rows finite, images with identical content and points give identical bits wherever they land in a batch, a
    resumed run touches nothing."""
    from conftest import GOLDEN, check_labels, cosine, rel_l2
    import torch
    from mermaid_classifier_amd import load_predictor
    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd.inference import params_from_torchscript
    from mermaid_classifier_amd.pipeline import BatchedExtractor
    from oracle import head_ref, pyspacer_ref
    from scipy.ndimage import zoom
    rng = np.random.default_rng(31)
    import os
    H, W, NIMG = 609, 696, int(os.environ.get("MMC_TEST_NIMG", "10000"))
    pool = []
    for i in range(8):                                     # 8 distinct image contents, smooth (image-like)
        base = rng.integers(0, 255, (H // 16 + 2, W // 16 + 2, 3)).astype(np.float32)
        pool.append(np.clip(zoom(base, (16, 16, 1), order=1)[:H, :W], 0, 255).astype(np.uint8))
    grid = [(int(H * (2 * i + 1) / 10), int(W * (2 * j + 1) / 10)) for i in range(5) for j in range(5)]
    ids = [f"{k:05d}" for k in range(NIMG)]
    shift = lambda k: 3 * ((k // 8) % 5)                   # noqa: E731  five point sets per content
    pts = {ids[k]: [(r + shift(k), c + shift(k)) for r, c in grid] for k in range(NIMG)}
    stored = {}

    def store(image_id, feats):
        stored[image_id] = np.asarray([pf.data for pf in feats.point_features], dtype=np.float32)

    bb = Backbone(synth_sd, device=0, max_batch=256)
    try:
        bx = BatchedExtractor(bb, batch_patches=1024)
        c = driver.process_source(source_id="9", images=pts, load_image=lambda i: pool[int(i) % 8], store_features=store, extractor=bx)
        assert (c.images_ok, c.images_failed, c.images_skipped) == (NIMG, 0, 0) and len(stored) == NIMG
        c2 = driver.process_source(source_id="9", images=pts, load_image=lambda i: pool[int(i) % 8], store_features=store,
                                   extractor=bx, existing=set(stored))
        assert (c2.images_ok, c2.images_skipped) == (0, NIMG)
    finally:
        bb.close()
    feats = np.concatenate([stored[i] for i in sorted(stored)])          # extract_reference_features.py:50-59 order
    assert feats.shape == (25 * NIMG, 1280) and np.isfinite(feats).all()
    for k in range(40, NIMG):                                            # same content + same points => same bits
        assert np.array_equal(stored[ids[k]], stored[ids[k - 40]]), k
    assert not np.array_equal(stored[ids[0]], stored[ids[8]])            # (different points do differ)
    chk = [0, 13]
    want = np.concatenate([pyspacer_ref.extract(oracle_net, pool[k % 8], pts[ids[k]]) for k in chk])
    got = np.concatenate([stored[ids[k]] for k in chk])
    assert cosine(got, want).min() >= 0.999 and rel_l2(got, want).max() < 2e-3
    pred = load_predictor(GOLDEN / "head108" / "model.pt", GOLDEN / "head108" / "model.json")
    prm = params_from_torchscript(torch.jit.load(str(GOLDEN / "head108" / "model.pt")))
    p_all = pred.predict_proba(feats)
    assert p_all.shape == (25 * NIMG, 108) and np.abs(p_all.sum(1) - 1).max() < 1e-5
    p_ref = head_ref.predict_proba(want, prm.weights, prm.biases, prm.a, prm.b, 1280)
    check_labels(pred.predict_proba(got), p_ref, dp_bound=2e-4, max_flips=1, what=f"50 oracle-checked patches of the {25 * NIMG}")
