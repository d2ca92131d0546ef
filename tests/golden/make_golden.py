#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run ONCE in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Head fixtures come from the REFERENCE ITSELF, imported from /root/reference
(``mermaid_classifier.pyspacer.inference.head`` / ``.loader`` /
``.torch_classifier`` and the reference's own test fixture
``tests/pyspacer/_calibrated_model_fixture.py``): inputs, parameters and the
outputs of its ``CalibratedHead`` / ``Predictor`` / sklearn ``predict_proba``.
``export_artifact`` itself cannot run here (sklearn pin 1.5.2 vs 1.7.2 installed,
and ``importlib.metadata.version('pyspacer')`` -- SURVEY 8c), so ``model.pt`` /
``model.json`` are written by the same calls it makes (export.py:52-57, 72-92).

Backbone fixtures are SELF-ORACLE (parity unpinned: pyspacer is absent): the
synthetic-weights BN statistics and the oracle's fp32 features on the seed-42
patches of scripts/build_feature_bucket.py:469-473 and on image-like patches.

Nothing from /root/reference is copied: only inputs/outputs (data) are stored.
"""

from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import efficientnet_b0_ref as bb  # noqa: E402
from oracle import pyspacer_ref  # noqa: E402


def backbone():
    torch.set_num_threads(os.cpu_count() or 1)
    sd = bb.make_synthetic_state_dict(seed=0, bn_stats=None)
    np.savez_compressed(HERE / "synth_bn_stats.npz", **bb.bn_stats_of(sd))
    net = bb.EfficientNetB0Ref(sd)
    noise = bb.synthetic_patches(8, seed=42)
    natural = bb.natural_patches(8, seed=7)
    f_noise = bb.patches_to_features(net, noise, batch_size=10)
    f_nat = bb.patches_to_features(net, natural, batch_size=10)
    # per-layer checksums on 2 patches (mean, mean|x|, max|x| of every tapped tensor)
    taps = {}
    net.extract_features(bb.transformation(noise[:2]), taps=taps)
    names = sorted(taps)
    sums = np.array([[taps[k].mean().item(), taps[k].abs().mean().item(), taps[k].abs().max().item()] for k in names],
                    dtype=np.float64)
    # config-1 geometry, scaled down: one synthetic image, the CSV's 5x5 grid scaled to it
    rng = np.random.default_rng(42)
    image = rng.integers(0, 255, (696, 928, 3), dtype=np.uint8)
    rowcols = [(r, c) for r in (0, 116, 348, 580, 695) for c in (0, 232, 464, 927)][:10]
    f_img = pyspacer_ref.extract(net, image, rowcols, batch_size=10)
    np.savez_compressed(HERE / "backbone_features.npz", noise8=f_noise, natural8=f_nat,
                        tap_names=np.array(names), tap_sums=sums,
                        image_rowcols=np.array(rowcols, dtype=np.int64), image_features=f_img)
    print("backbone fixtures written", f_noise.shape, f_nat.shape, f_img.shape)


def backbone_b4():
    """BASELINE.json configs[4]: EfficientNet-B4 (width 1.4, depth 1.8) on 224x224 patches.  Not in the
    reference: the oracle is the same restatement over the scaled stage table; self-oracle fixtures."""
    torch.set_num_threads(os.cpu_count() or 1)
    sd = bb.make_synthetic_state_dict(seed=0, bn_stats=None, arch="b4")
    np.savez_compressed(HERE / "synth_bn_stats_b4.npz", **{k: v.astype(np.float16) for k, v in bb.bn_stats_of(sd).items()})
    # the committed statistics are fp16-rounded (half the file); rebuild the weights from them so that the
    # fixture below is computed with exactly what every later run reconstructs
    stats = {k: v.astype(np.float32) for k, v in np.load(HERE / "synth_bn_stats_b4.npz").items()}
    sd = bb.make_synthetic_state_dict(seed=0, bn_stats=stats, arch="b4")
    net = bb.EfficientNetB0Ref(sd, arch="b4")
    noise = bb.synthetic_patches(4, seed=42)
    natural = bb.natural_patches(4, seed=7)
    np.savez_compressed(HERE / "backbone_b4_features.npz", noise4=bb.patches_to_features(net, noise),
                        natural4=bb.patches_to_features(net, natural))
    print("b4 fixtures written")


def strip_debug_pkl(path: Path) -> int:
    """Drop the `*.debug_pkl` members torch.jit.save adds to a TorchScript archive.  They hold the source ranges -- and with
    them the TEXT -- of the scripted module's Python source (here the reference's inference/head.py), which must not travel
    with the fixtures; torch.jit.load does not need them (it only loses file:line in error messages).  The other members are
    copied as they are: STORED, data records padded to 64-byte offsets like torch's own writer does.  Returns members removed."""
    import zipfile
    src = zipfile.ZipFile(path)
    infos = src.infolist()
    keep = [i for i in infos if not i.filename.endswith(".debug_pkl")]
    if len(keep) == len(infos):
        return 0
    tmp = Path(str(path) + ".tmp")
    with zipfile.ZipFile(tmp, "w", compression=zipfile.ZIP_STORED) as dst:
        for i in keep:
            zi = zipfile.ZipInfo(i.filename, date_time=(1980, 1, 1, 0, 0, 0))
            zi.compress_type = zipfile.ZIP_STORED
            # pad the local header's extra field so the payload starts on a 64-byte boundary (what PyTorchStreamWriter does)
            base = dst.fp.tell() + 30 + len(i.filename.encode())
            pad = (-(base + 4)) % 64
            zi.extra = b"FB" + pad.to_bytes(2, "little") + b"Z" * pad
            dst.writestr(zi, src.read(i.filename))
    src.close()
    tmp.replace(path)
    return len(infos) - len(keep)


def head():
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, "/root/reference/tests")
    from mermaid_classifier.pyspacer.inference import SCHEMA_VERSION, TASK_NAME  # type: ignore
    from mermaid_classifier.pyspacer.inference.head import build_calibrated_head  # type: ignore
    from mermaid_classifier.pyspacer.inference.loader import load_predictor  # type: ignore
    from mermaid_classifier.pyspacer.torch_classifier import TorchMLPClassifier  # type: ignore
    from pyspacer._calibrated_model_fixture import make_calibrated_model  # type: ignore
    from sklearn.calibration import CalibratedClassifierCV, _fit_calibrator
    import sklearn

    def export(model, out_dir: Path, input_dim: int):
        out_dir.mkdir(parents=True, exist_ok=True)
        h = build_calibrated_head(model)
        h.eval()
        frozen = torch.jit.freeze(torch.jit.script(h))
        torch.jit.save(frozen, str(out_dir / "model.pt"))
        strip_debug_pkl(out_dir / "model.pt")     # no reference source text inside the fixture
        manifest = {
            "schema_version": SCHEMA_VERSION, "task": TASK_NAME,
            "classes": model.classes_.tolist(), "input_dim": int(input_dim),
            "config": {"patch_size": 224},
            "trained_with": {"torch": torch.__version__, "sklearn": sklearn.__version__, "pyspacer": "0.14.0"},
        }
        (out_dir / "model.json").write_text(json.dumps(manifest, indent=2))
        return h

    def dump(tag: str, model, h, X, out_dir: Path):
        with torch.no_grad():
            eager = h(torch.from_numpy(X.astype(np.float32))).numpy()
        pred = load_predictor(out_dir / "model.pt", out_dir / "model.json")
        got = pred.predict_proba(X)
        sk = model.predict_proba(X)
        print(tag, "max|head-sklearn|", np.abs(eager - sk).max(), "max|predictor-eager|", np.abs(got - eager).max())
        arrs = {"X": X.astype(np.float32), "proba_head_f32": eager.astype(np.float32),
                "proba_predictor_f64": got, "proba_sklearn_f64": sk,
                "a": h.a.numpy(), "b": h.b.numpy(), "n_layers": np.array(len(h.linears))}
        for i, lin in enumerate(h.linears):
            arrs[f"W{i}"] = lin.weight.detach().numpy()
            arrs[f"b{i}"] = lin.bias.detach().numpy()
        return arrs

    # (1) the reference's own small test fixture: 8 -> 16 -> 5
    model, X = make_calibrated_model()
    d = HERE / "head_fixture"
    h = export(model, d, X.shape[1])
    np.savez_compressed(HERE / "head_fixture_io.npz", **dump("fixture", model, h, X, d))

    # (2) production-shaped head: 1280 -> (500,300,100) -> 108 (trainer.py:118-123), seed 0
    rng = np.random.default_rng(0)
    k, nf, ns = 108, 1280, 6000
    classes = np.array([f"ba{i:03d}::gf{i:03d}" for i in range(k)])
    base = rng.normal(0.4, 0.35, size=(1, nf))
    centers = np.abs(base + 0.10 * rng.normal(size=(k, nf))).astype(np.float32)
    yi = rng.integers(0, k, size=ns)
    Xb = np.abs(centers[yi] + rng.normal(0, 0.25, size=(ns, nf))).astype(np.float32)
    y = classes[yi]
    clf = TorchMLPClassifier(hidden_layer_sizes=(500, 300, 100), learning_rate_init=1e-3, random_state=0)
    for _ in range(12):
        clf.partial_fit(Xb, y, classes=classes.tolist())
    preds = clf.predict_proba(Xb)
    inner = _fit_calibrator(clf, preds, y, clf.classes_, method="sigmoid")
    wrapper = CalibratedClassifierCV(clf, cv="prefit")
    wrapper.calibrated_classifiers_ = [inner]
    wrapper.classes_ = clf.classes_
    d = HERE / "head108"
    h = export(wrapper, d, nf)
    # evaluation inputs: held-out samples of the same family (hard + easy rows)
    yi2 = rng.integers(0, k, size=256)
    Xe = np.abs(centers[yi2] + rng.normal(0, 0.45, size=(256, nf))).astype(np.float32)
    arrs = dump("head108", wrapper, h, Xe, d)
    # weights live in model.pt; keep the io file small
    small = {k_: v for k_, v in arrs.items() if not (k_.startswith("W") or (k_.startswith("b") and k_[1:].isdigit()))}
    np.savez_compressed(HERE / "head108_io.npz", **small)
    acc = (np.argmax(arrs["proba_head_f32"], 1) == yi2).mean()
    print("head108 held-out accuracy", acc)


def trainer():
    """the reference's own TorchMLPClassifier (imported from /root/reference) trained for three
    partial_fit passes on a seeded dataset -- inputs, initial and final parameters, loss curve, probabilities."""
    sys.path.insert(0, "/root/reference")
    from mermaid_classifier.pyspacer.torch_classifier import TorchMLPClassifier, _MLPModule  # type: ignore
    rng = np.random.default_rng(5)
    k, nf, n = 7, 64, 730                      # 730 = 3 x 200 + a ragged mini-batch of 130
    classes = np.array([f"c{i}" for i in range(k)])
    centers = rng.normal(0, 1.0, size=(k, nf)).astype(np.float32)
    yi = rng.integers(0, k, size=n)
    X = (centers[yi] + rng.normal(0, 1.5, size=(n, nf))).astype(np.float32)
    y = classes[yi]
    cw = {c: float(w) for c, w in zip(classes, [1.0, 0.5, 2.0, 1.0, 0.0, 3.0, 1.5])}
    out = {"X": X, "y_idx": yi.astype(np.int64), "classes": classes, "class_weight": np.array([cw[c] for c in classes], np.float32)}
    for tag, weight in (("plain", None), ("weighted", cw)):
        torch.manual_seed(0)
        init = _MLPModule(nf, (48, 32), k)
        clf = TorchMLPClassifier(hidden_layer_sizes=(48, 32), learning_rate_init=1e-3, alpha=1e-3, random_state=0, class_weight=weight)
        for _ in range(3):
            clf.partial_fit(X, y, classes=classes.tolist())
        for i, (m0, m1) in enumerate(zip(init.linears, clf._module.linears)):
            if tag == "plain":
                out[f"W{i}_init"] = m0.weight.detach().numpy().copy()
                out[f"b{i}_init"] = m0.bias.detach().numpy().copy()
            out[f"W{i}_{tag}"] = m1.weight.detach().numpy().copy()
            out[f"b{i}_{tag}"] = m1.bias.detach().numpy().copy()
        out[f"loss_curve_{tag}"] = np.array(clf.loss_curve_, dtype=np.float64)
        out[f"proba_{tag}"] = clf.predict_proba(X[:64])
        print("trainer", tag, "loss curve", clf.loss_curve_, "acc", (clf.predict(X) == y).mean())
    np.savez_compressed(HERE / "trainer_fixture.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["backbone", "head"]
    if "backbone" in which:
        backbone()
    if "backbone_b4" in which:
        backbone_b4()
    if "head" in which:
        head()
    if "trainer" in which:
        trainer()
    if "strip" in which:      # post-process archives that were written before strip_debug_pkl existed
        for pt in sorted(HERE.glob("*/model.pt")):
            print(pt, "removed", strip_debug_pkl(pt), "debug_pkl member(s)")
