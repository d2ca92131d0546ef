#!/usr/bin/env python3
"""GPU box: samples/s of TorchMLPClassifier.partial_fit on the MI355X (production shape 1280 -> 500 -> 300 -> 100 -> 108,
mini-batches of 200 = the reference's "auto") next to the numpy oracle of the same arithmetic on the host cores."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mermaid_classifier_amd.torch_classifier import TorchMLPClassifier
from oracle.mlp_train_ref import MLPTrainRef
rng = np.random.default_rng(0)
k, nf, n = 108, 1280, 20000
yi = rng.integers(0, k, size=n)
X = np.abs(rng.normal(0.4, 0.4, size=(n, nf))).astype(np.float32)
clf = TorchMLPClassifier(hidden_layer_sizes=(500, 300, 100), learning_rate_init=1e-4, random_state=0)
clf.partial_fit(X, yi, classes=list(range(k)))
t0 = time.perf_counter()
for _ in range(3):
    clf.partial_fit(X, yi)
dt = (time.perf_counter() - t0) / 3
print(f"HIP partial_fit: {n/dt:.0f} samples/s ({dt*1e3:.1f} ms per pass of {n}, {n//200} Adam steps, incl. host shuffle + H2D of {X.nbytes/1e6:.0f} MB)")
clf2 = TorchMLPClassifier(hidden_layer_sizes=(500, 300, 100), random_state=0)
clf2.classes_, clf2.n_features_in_ = np.arange(k), nf
ref = MLPTrainRef(*clf2._initial_parameters(), lr=1e-4)
t0 = time.perf_counter()
ref.partial_fit(X[:4000], yi[:4000], "auto")
dt2 = time.perf_counter() - t0
print(f"numpy oracle on the host: {4000/dt2:.0f} samples/s")
