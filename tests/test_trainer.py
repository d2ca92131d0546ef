"""MLP classifier training.  The golden fixture was produced by the REFERENCE's own
TorchMLPClassifier (tests/golden/make_golden.py ``trainer``, importing /root/reference): three partial_fit passes,
730 samples (mini-batches 200,200,200,130), hidden (48,32), 7 classes, alpha 1e-3, random_state 0, with and without
class weights (one of them 0).

CPU: the numpy oracle against the fixture; the host class's initial weights and shuffle against the fixture.
GPU (-m gpu): the HIP trainer through the C ABI against the fixture and the oracle.
Tolerance: 12 Adam steps in fp32 with different summation orders -- weights within 2e-5 absolute (they are O(0.1)),
loss curve within 1e-5, probabilities within 1e-5."""

import pickle

import numpy as np
import pytest

from conftest import GOLDEN

W_TOL, LOSS_TOL, P_TOL = 2e-5, 1e-5, 1e-5


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(GOLDEN / "trainer_fixture.npz"))


def _init(fx):
    return [fx[f"W{i}_init"] for i in range(3)], [fx[f"b{i}_init"] for i in range(3)]


@pytest.mark.parametrize("tag", ["plain", "weighted"])
def test_oracle_matches_reference_training(fx, tag):
    from oracle.mlp_train_ref import MLPTrainRef
    ws, bs = _init(fx)
    ref = MLPTrainRef(ws, bs, lr=1e-3, alpha=1e-3, class_weight=fx["class_weight"] if tag == "weighted" else None)
    curve = [ref.partial_fit(fx["X"], fx["y_idx"], "auto", shuffle=True, random_state=0) for _ in range(3)]
    np.testing.assert_allclose(curve, fx[f"loss_curve_{tag}"], atol=LOSS_TOL)
    for i in range(3):
        assert np.abs(ref.W[i] - fx[f"W{i}_{tag}"]).max() < W_TOL
        assert np.abs(ref.b[i] - fx[f"b{i}_{tag}"]).max() < W_TOL
    assert np.abs(ref.predict_proba(fx["X"][:64]) - fx[f"proba_{tag}"]).max() < P_TOL


def test_host_class_reproduces_reference_initialisation_and_errors(fx):
    """Glorot init under torch.manual_seed(random_state) is drawn in the reference's order (torch_classifier.py:53-76,
    176-184); constructor/label errors are the reference's (no GPU needed for any of this)."""
    from mermaid_classifier_amd.torch_classifier import TorchMLPClassifier
    clf = TorchMLPClassifier(hidden_layer_sizes=(48, 32), random_state=0)
    clf.classes_ = fx["classes"]
    clf.n_features_in_ = 64
    ws, bs = clf._initial_parameters()
    for i in range(3):
        assert np.array_equal(ws[i], fx[f"W{i}_init"]) and np.array_equal(bs[i], fx[f"b{i}_init"])
    assert clf._resolve_batch_size(730) == 200 and clf._resolve_batch_size(50) == 50
    assert np.array_equal(clf._labels_to_indices(np.array(["c3", "c0"])), [3, 0])
    with pytest.raises(ValueError, match="not in classes_"):
        clf._labels_to_indices(np.array(["zz"]))
    with pytest.raises(ValueError):
        TorchMLPClassifier(activation="tanh")
    with pytest.raises(ValueError):
        TorchMLPClassifier(solver="sgd")
    with pytest.raises(RuntimeError, match="not fitted"):
        clf.predict_proba(np.zeros((1, 64), np.float32))
    clf.class_weight = {"c0": 1.0}
    with pytest.raises(ValueError, match="missing weights"):
        clf._build_class_weight_vector()
    assert clf.get_params()["hidden_layer_sizes"] == (48, 32)
    with pytest.raises(ValueError):
        clf.set_params(bogus=1)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "weighted"])
def test_hip_trainer_matches_reference_training(fx, tag):
    from mermaid_classifier_amd.torch_classifier import TorchMLPClassifier
    classes = fx["classes"]
    cw = {c: float(w) for c, w in zip(classes, fx["class_weight"])} if tag == "weighted" else None
    clf = TorchMLPClassifier(hidden_layer_sizes=(48, 32), learning_rate_init=1e-3, alpha=1e-3, random_state=0, class_weight=cw)
    y = classes[fx["y_idx"]]
    for _ in range(3):
        clf.partial_fit(fx["X"], y, classes=classes.tolist())
    assert clf.n_iter_ == 3 and np.array_equal(clf.classes_, classes)
    print(tag, "loss curve", clf.loss_curve_, "reference", fx[f"loss_curve_{tag}"])
    np.testing.assert_allclose(clf.loss_curve_, fx[f"loss_curve_{tag}"], atol=LOSS_TOL)
    ws, bs = clf.parameters()
    for i in range(3):
        dw, db = np.abs(ws[i] - fx[f"W{i}_{tag}"]).max(), np.abs(bs[i] - fx[f"b{i}_{tag}"]).max()
        print(f"layer {i}: max|dW| {dw:.2e} max|db| {db:.2e}")
        assert dw < W_TOL and db < W_TOL
    proba = clf.predict_proba(fx["X"][:64])
    assert proba.dtype == np.float64 and np.abs(proba - fx[f"proba_{tag}"]).max() < P_TOL
    np.testing.assert_allclose(proba.sum(axis=1), 1.0, atol=1e-12)
    assert np.array_equal(clf.predict(fx["X"][:64]), classes[fx[f"proba_{tag}"].argmax(1)])
    # the module view the export path reads (inference/head.py build_calibrated_head)
    lin = clf._module.linears
    assert len(lin) == 3 and np.array_equal(lin[0].weight.detach().numpy(), ws[0])
    # pickle round trip keeps parameters AND optimizer state: a further pass gives identical bits on both copies
    twin = pickle.loads(pickle.dumps(clf))
    clf.partial_fit(fx["X"], y)
    twin.partial_fit(fx["X"], y)
    for a, b in zip(clf.parameters()[0], twin.parameters()[0]):
        assert np.array_equal(a, b)
    assert clf.loss_curve_ == twin.loss_curve_
    with pytest.raises(ValueError):
        clf.partial_fit(np.zeros((4, 63), np.float32), y[:4])
    with pytest.raises(ValueError, match="not in classes_"):
        clf.partial_fit(fx["X"][:4], np.array(["c0", "c1", "nope", "c2"]))


@pytest.mark.gpu
def test_hip_trainer_production_shape_against_oracle():
    """1280 -> 500 -> 300 -> 100 -> 108 (trainer.py:118-123), explicit batch size, ragged last mini-batch."""
    from mermaid_classifier_amd.torch_classifier import TorchMLPClassifier
    from oracle.mlp_train_ref import MLPTrainRef
    rng = np.random.default_rng(11)
    k, nf, n = 108, 1280, 1100
    centers = np.abs(rng.normal(0.4, 0.35, size=(k, nf))).astype(np.float32)
    yi = rng.integers(0, k, size=n)
    X = np.abs(centers[yi] + rng.normal(0, 0.25, size=(n, nf))).astype(np.float32)
    clf = TorchMLPClassifier(hidden_layer_sizes=(500, 300, 100), learning_rate_init=1e-4, batch_size=256, random_state=3)
    clf.classes_ = np.arange(k)
    clf.n_features_in_ = nf
    w0, b0 = clf._initial_parameters()
    del clf.classes_, clf.n_features_in_
    ref = MLPTrainRef(w0, b0, lr=1e-4, alpha=1e-4)
    want = [ref.partial_fit(X, yi, 256, random_state=3) for _ in range(2)]
    for _ in range(2):
        clf.partial_fit(X, yi, classes=list(range(k)))
    print("loss", clf.loss_curve_, "oracle", want)
    np.testing.assert_allclose(clf.loss_curve_, want, atol=2e-5)
    # Adam normalises every element's step to ~lr whatever the gradient's size, so where a gradient is at rounding-noise
    # level (inputs that are almost always 0 behind a ReLU) its SIGN -- hence a whole step of 1e-4 -- depends on fp32
    # summation order: any two fp32 implementations disagree on a handful of such elements.  Bulk agreement is the
    # check: 99.9 % of the elements within W_TOL, none further apart than two steps.
    for a, b in zip(clf.parameters()[0], ref.W):
        d = np.abs(a - b)
        print(f"{a.shape}: mean {d.mean():.2e} p99.9 {np.quantile(d, 0.999):.2e} max {d.max():.2e}")
        assert np.quantile(d, 0.999) < W_TOL and d.max() < 2e-4
    assert np.abs(clf.predict_proba(X[:50]) - ref.predict_proba(X[:50])).max() < P_TOL


@pytest.mark.gpu
def test_hip_trainer_odd_feature_width_takes_the_host_shuffle_path():
    """A feature width that is not a multiple of 4 cannot use the device-side gather (float4 rows): the host applies the
    visiting order instead (mmc_trainer_partial_fit); both paths must follow the oracle."""
    from mermaid_classifier_amd.torch_classifier import TorchMLPClassifier
    from oracle.mlp_train_ref import MLPTrainRef
    rng = np.random.default_rng(2)
    k, nf, n = 5, 10, 333
    yi = rng.integers(0, k, size=n)
    X = (rng.normal(0, 1, size=(k, nf))[yi] + rng.normal(0, 1, size=(n, nf))).astype(np.float32)
    clf = TorchMLPClassifier(hidden_layer_sizes=(16,), batch_size=50, random_state=1)
    clf.classes_, clf.n_features_in_ = np.arange(k), nf
    w0, b0 = clf._initial_parameters()
    del clf.classes_, clf.n_features_in_
    ref = MLPTrainRef(w0, b0)
    want = [ref.partial_fit(X, yi, 50, random_state=1) for _ in range(2)]
    for _ in range(2):
        clf.partial_fit(X, yi, classes=list(range(k)))
    np.testing.assert_allclose(clf.loss_curve_, want, atol=LOSS_TOL)
    for a, b in zip(clf.parameters()[0], ref.W):
        assert np.abs(a - b).max() < W_TOL
