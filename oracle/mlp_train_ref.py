"""CPU oracle for the MLP classifier's training step.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  Plain-numpy float32 restatement of the arithmetic of
``TorchMLPClassifier.partial_fit`` (reference mermaid_classifier/pyspacer/torch_classifier.py:226-303):

* visiting order: ``default_rng(random_state).shuffle(arange(n))`` re-seeded on every call (:139-147, :251-254);
* per mini-batch of ``min(batch_size or 200, n)`` rows (:133-137, :263-268): logits of the Linear/ReLU stack (:70-76),
  ``F.cross_entropy(logits, y, weight=w)`` = sum_i w[y_i] nll_i / sum_i w[y_i] (:270-278), plus
  ``(0.5 * alpha / mb) * sum(W**2)`` over the weights only (:214-224, :279-285), backward, ``torch.optim.Adam`` step
  (:186-192: lerp / addcmul / addcdiv order, bias corrections from the global step count);
* ``loss_curve_`` entry = sum(loss_i * mb_i) / n (:293-300).

PINNED: tests/test_oracle.py checks it against tests/golden/trainer_fixture.npz, produced by importing and running the
reference's own TorchMLPClassifier (tests/golden/make_golden.py ``trainer``): weights after three passes, loss curve,
predict_proba.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


class MLPTrainRef:
    def __init__(self, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], lr=1e-3, beta1=0.9, beta2=0.999,
                 eps=1e-8, alpha=1e-4, class_weight: Optional[np.ndarray] = None):
        self.W = [np.array(w, dtype=F32) for w in weights]
        self.b = [np.array(v, dtype=F32) for v in biases]
        self.mW = [np.zeros_like(w) for w in self.W]
        self.vW = [np.zeros_like(w) for w in self.W]
        self.mb = [np.zeros_like(v) for v in self.b]
        self.vb = [np.zeros_like(v) for v in self.b]
        self.lr, self.beta1, self.beta2, self.eps, self.alpha = lr, beta1, beta2, eps, alpha
        self.cw = None if class_weight is None else np.asarray(class_weight, dtype=F32)
        self.t = 0

    def logits(self, X: np.ndarray) -> np.ndarray:
        h = np.asarray(X, dtype=F32)
        for i, (w, b) in enumerate(zip(self.W, self.b)):
            h = h @ w.T + b
            if i < len(self.W) - 1:
                h = np.maximum(h, F32(0))
        return h

    def step(self, X: np.ndarray, y: np.ndarray) -> float:
        """One mini-batch: returns the regularised loss before the update."""
        mb = X.shape[0]
        hs = [np.asarray(X, dtype=F32)]
        for i, (w, b) in enumerate(zip(self.W, self.b)):
            z = hs[-1] @ w.T + b
            hs.append(np.maximum(z, F32(0)) if i < len(self.W) - 1 else z)
        z = hs[-1]
        zs = z - z.max(axis=1, keepdims=True)
        lse = np.log(np.exp(zs).sum(axis=1, dtype=F32))
        logp = zs - lse[:, None]
        w_i = np.ones(mb, F32) if self.cw is None else self.cw[y]
        wsum = F32(w_i.sum(dtype=np.float64))
        data = F32((w_i * -logp[np.arange(mb), y]).sum(dtype=np.float64)) / wsum
        reg = F32(0.5 * self.alpha / mb) * F32(sum(float((w.astype(np.float64) ** 2).sum()) for w in self.W))
        dz = np.exp(logp)
        dz[np.arange(mb), y] -= F32(1)
        dz *= (w_i / wsum)[:, None]
        gW, gb = [None] * len(self.W), [None] * len(self.W)
        for l in range(len(self.W) - 1, -1, -1):
            gW[l] = dz.T @ hs[l] + F32(self.alpha / mb) * self.W[l]
            gb[l] = dz.sum(axis=0, dtype=F32)
            if l > 0:
                dz = (dz @ self.W[l]) * (hs[l] > 0)
        self.t += 1
        bc1 = 1.0 - self.beta1 ** self.t
        bc2 = 1.0 - self.beta2 ** self.t
        step_size, bc2_sqrt = F32(self.lr / bc1), F32(np.sqrt(bc2))
        for P, G, M, V in ((self.W, gW, self.mW, self.vW), (self.b, gb, self.mb, self.vb)):
            for l in range(len(P)):
                g = G[l].astype(F32)
                M[l] = M[l] + F32(1 - self.beta1) * (g - M[l])
                V[l] = V[l] * F32(self.beta2) + F32(1 - self.beta2) * g * g
                P[l] = P[l] - step_size * (M[l] / (np.sqrt(V[l]) / bc2_sqrt + F32(self.eps)))
        return float(data + reg)

    def partial_fit(self, X: np.ndarray, y_idx: np.ndarray, batch_size, shuffle=True, random_state: Optional[int] = 0) -> float:
        n = X.shape[0]
        mb = min(200, n) if batch_size == "auto" else min(int(batch_size), n)
        order = np.arange(n)
        if shuffle:
            np.random.default_rng(int(random_state)).shuffle(order)
        Xo, yo = np.asarray(X, dtype=F32)[order], np.asarray(y_idx)[order]
        tot = 0.0
        for s in range(0, n, mb):
            e = min(s + mb, n)
            tot += self.step(Xo[s:e], yo[s:e]) * (e - s)
        return tot / n

    def predict_proba(self, X: np.ndarray) -> np.ndarray:
        z = self.logits(X)
        z = z - z.max(axis=1, keepdims=True)
        e = np.exp(z, dtype=F32)
        p = (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(np.float64)
        return p / p.sum(axis=1, keepdims=True)
