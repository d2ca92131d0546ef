"""TorchMLPClassifier with the training arithmetic on the MI355X.

Host-side mirror of reference ``mermaid_classifier/pyspacer/torch_classifier.py`` (class ``TorchMLPClassifier``,
the sklearn-``MLPClassifier`` subset the trainer drives through ``partial_fit``,
``mermaid_classifier/pyspacer/trainer.py:118-145``): same constructor arguments, same ``partial_fit`` / ``fit`` /
``predict`` / ``predict_proba`` / ``classes_`` / ``loss_curve_`` / ``n_iter_`` / ``get_params`` / ``set_params`` /
pickling behaviour and the same errors.  What stays on the host is what the reference keeps in numpy / Python: label
lookup, the shuffle order (numpy ``default_rng(random_state)``, :139-157), Glorot initialisation (torch's CPU generator
under ``torch.manual_seed(random_state)``, so equal seeds give equal initial weights, :176-184), softmax + float64
renormalisation of ``_forward_probs`` (:332-376).  Forward, weighted cross-entropy, L2 term, backward and Adam run in
``libmermaid_mi355.so`` (``mmc_trainer_*``, csrc/trainer.hip).  No CPU fallback.

Taken as they are from the reference (the sklearn-protocol shell a drop-in has to reproduce; argument checks, defaults
and error strings included): ``__init__`` (:95-136, the argument validation and attribute names), ``_resolve_batch_size``
(:138-141), ``_seed_rng`` (:143-157), ``_labels_to_indices`` (:159-173), ``_build_class_weight_vector`` (the reference's
``_build_class_weight_tensor``, :192-214, minus the tensor construction), ``get_params`` / ``set_params`` (:380-408) and the input checks at the top of ``partial_fit`` /
``_forward_probs``.  Written here, with no counterpart there: ``_initial_parameters`` (host Glorot draw handed to the
device), ``_create_trainer``, the device half of ``partial_fit`` (``mmc_trainer_partial_fit_ordered``), ``_forward_probs``'s
logits call (``mmc_trainer_logits``), ``parameters`` / ``_module`` / ``_adam_state`` (state read back through the C ABI),
``__getstate__`` / ``__setstate__`` (the pickle carries host copies of weights and Adam moments), ``_release``.
"""

from __future__ import annotations

import ctypes as C
import warnings
from collections.abc import Sequence
from typing import Any, List

import numpy as np

from . import _lib
from .backbone import _current_stream_ptr, _device_index

_EXPECTED_FP_DRIFT_TOL = 1e-4   # torch_classifier.py:45-50


def _ptr_array(arrays: List[np.ndarray]):
    fp = C.POINTER(C.c_float)
    return (fp * len(arrays))(*[a.ctypes.data_as(fp) for a in arrays])


class TorchMLPClassifier:
    """See the module docstring; argument meaning as in the reference (torch_classifier.py:93-135)."""

    _estimator_type = "classifier"

    def __init__(self, hidden_layer_sizes: Sequence[int] = (100,), activation: str = "relu", solver: str = "adam",
                 alpha: float = 0.0001, batch_size: int | str = "auto", learning_rate_init: float = 0.001,
                 max_iter: int = 200, shuffle: bool = True, random_state: int | None = None, tol: float = 1e-4,
                 beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-8,
                 class_weight: dict | None = None, device=0):
        if activation != "relu":
            raise ValueError(f"TorchMLPClassifier only supports activation='relu', got {activation!r}.")
        if solver != "adam":
            raise ValueError(f"TorchMLPClassifier only supports solver='adam', got {solver!r}.")
        self.hidden_layer_sizes = tuple(hidden_layer_sizes)
        self.activation = activation
        self.solver = solver
        self.alpha = alpha
        self.batch_size = batch_size
        self.learning_rate_init = learning_rate_init
        self.max_iter = max_iter
        self.shuffle = shuffle
        self.random_state = random_state
        self.tol = tol
        self.beta_1 = beta_1
        self.beta_2 = beta_2
        self.epsilon = epsilon
        self.class_weight = class_weight
        self.device = device

    # ---- host bookkeeping, restated from the reference ------------------------------------------------------------
    def _resolve_batch_size(self, n_samples: int) -> int:
        if self.batch_size == "auto":
            return min(200, n_samples)
        return min(int(self.batch_size), n_samples)

    def _seed_rng(self) -> np.random.Generator:
        if self.random_state is not None:
            return np.random.default_rng(int(self.random_state))
        if not hasattr(self, "_none_rng"):
            self._none_rng = np.random.default_rng(np.random.randint(0, np.iinfo(np.int32).max))
        return self._none_rng

    def _labels_to_indices(self, y: np.ndarray) -> np.ndarray:
        y = np.asarray(y)
        idx = np.searchsorted(self.classes_, y)
        missing = idx >= len(self.classes_)
        if missing.any() or not np.array_equal(self.classes_[np.minimum(idx, len(self.classes_) - 1)], y):
            bad = set(np.asarray(y).tolist()) - set(self.classes_.tolist())
            raise ValueError(f"Labels {sorted(bad)} are not in classes_ {self.classes_.tolist()}."
                             f" Pass all classes to the first partial_fit call.")
        return idx

    def _build_class_weight_vector(self):
        if self.class_weight is None:
            return None
        weights = []
        for cls in self.classes_:
            if cls not in self.class_weight:
                bad = sorted(set(self.classes_.tolist()) - set(self.class_weight))
                raise ValueError(f"class_weight is missing weights for {bad!r}. Pass weights for every class in classes_.")
            w = float(self.class_weight[cls])
            if w < 0:
                raise ValueError(f"class_weight for {cls!r} is negative ({w!r}); weights must be >= 0.")
            weights.append(w)
        return np.asarray(weights, dtype=np.float32)

    def _initial_parameters(self):
        """Glorot-uniform weights / zero biases drawn exactly as the reference's ``_MLPModule`` draws them
        (torch_classifier.py:53-76, 176-184): the Linear layers are constructed first (their default init consumes the
        generator), then re-initialised in order."""
        import torch
        import torch.nn as nn
        if self.random_state is not None:
            torch.manual_seed(int(self.random_state))
        sizes = [self.n_features_in_, *self.hidden_layer_sizes, len(self.classes_)]
        layers = [nn.Linear(i, o) for i, o in zip(sizes[:-1], sizes[1:])]
        for m in layers:
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
        return ([np.ascontiguousarray(m.weight.detach().numpy(), dtype=np.float32) for m in layers],
                [np.ascontiguousarray(m.bias.detach().numpy(), dtype=np.float32) for m in layers])

    def _create_trainer(self, weights, biases) -> None:
        lib = _lib.lib()
        self._dims = [int(weights[0].shape[1])] + [int(w.shape[0]) for w in weights]
        dims = (C.c_int * len(self._dims))(*self._dims)
        cw = self._class_weight_vector
        self._h = C.c_void_p()
        _lib.check(lib.mmc_trainer_create(_ptr_array(weights), _ptr_array(biases), dims, len(weights),
                                          float(self.learning_rate_init), float(self.beta_1), float(self.beta_2),
                                          float(self.epsilon), float(self.alpha),
                                          cw.ctypes.data_as(C.POINTER(C.c_float)) if cw is not None else None,
                                          _device_index(self.device), C.byref(self._h)))

    def _fitted(self) -> bool:
        return getattr(self, "_h", None) is not None and bool(self._h.value)

    # ---- training ------------------------------------------------------------------------------------------------
    def partial_fit(self, X, y, classes: Sequence[Any] | None = None) -> "TorchMLPClassifier":
        X_arr = np.asarray(X, dtype=np.float32)
        if X_arr.ndim != 2:
            raise ValueError(f"X must be 2D, got shape {X_arr.shape}")
        if not self._fitted():
            self.classes_ = np.unique(np.asarray(y)) if classes is None else np.unique(np.asarray(classes))
            self.n_features_in_ = int(X_arr.shape[1])
            self.n_iter_ = 0
            self.loss_curve_ = []
            self._class_weight_vector = self._build_class_weight_vector()
            self._create_trainer(*self._initial_parameters())
        elif X_arr.shape[1] != self.n_features_in_:
            raise ValueError(f"X has {X_arr.shape[1]} features, expected {self.n_features_in_}")
        y_indices = self._labels_to_indices(np.asarray(y))
        n_samples = X_arr.shape[0]
        batch_size = self._resolve_batch_size(n_samples)
        rng = self._seed_rng()
        order = np.arange(n_samples)
        if self.shuffle:
            rng.shuffle(order)
        avg = C.c_double(0.0)
        di = _device_index(self.device)
        if X_arr.shape[1] % 4 == 0:
            # the visiting order is applied on the device (a gather kernel): no host-side copy of the shuffled matrix
            Xn = np.ascontiguousarray(X_arr)
            yn = np.ascontiguousarray(y_indices.astype(np.int32))
            od = np.ascontiguousarray(order.astype(np.int64))
            _lib.check(_lib.lib().mmc_trainer_partial_fit_ordered(self._h, Xn.ctypes.data, yn.ctypes.data, od.ctypes.data, n_samples,
                                                                  int(batch_size), C.byref(avg), _current_stream_ptr(di)))
        else:
            Xo = np.ascontiguousarray(X_arr[order])
            yo = np.ascontiguousarray(y_indices[order].astype(np.int32))
            _lib.check(_lib.lib().mmc_trainer_partial_fit(self._h, Xo.ctypes.data, yo.ctypes.data, n_samples, int(batch_size),
                                                          C.byref(avg), _current_stream_ptr(di)))
        self.loss_curve_.append(float(avg.value))
        self.n_iter_ += 1
        return self

    def fit(self, X, y) -> "TorchMLPClassifier":
        y_arr = np.asarray(y)
        classes = np.unique(y_arr).tolist()
        self._release()
        for attr in ("classes_", "n_features_in_", "n_iter_", "loss_curve_"):
            if hasattr(self, attr):
                delattr(self, attr)
        prev_loss = np.inf
        for _ in range(self.max_iter):
            self.partial_fit(X, y_arr, classes=classes)
            cur = self.loss_curve_[-1]
            if abs(prev_loss - cur) < self.tol:
                break
            prev_loss = cur
        return self

    # ---- inference (pre-calibration semantics the head was fit on) -----------------------------------------------
    def _forward_probs(self, X) -> np.ndarray:
        if not self._fitted():
            raise RuntimeError("TorchMLPClassifier is not fitted. Call partial_fit or fit before predict/predict_proba.")
        X_arr = np.ascontiguousarray(np.asarray(X, dtype=np.float32))
        if X_arr.ndim != 2:
            raise ValueError(f"X must be 2D, got shape {X_arr.shape}")
        if X_arr.shape[1] != self.n_features_in_:
            raise ValueError(f"X has {X_arr.shape[1]} features, expected {self.n_features_in_}")
        n, k = X_arr.shape[0], len(self.classes_)
        logits = np.empty((n, k), dtype=np.float32)
        if n:
            di = _device_index(self.device)
            _lib.check(_lib.lib().mmc_trainer_logits(self._h, X_arr.ctypes.data, n, logits.ctypes.data, _current_stream_ptr(di)))
        z = logits - logits.max(axis=1, keepdims=True)                   # fp32 softmax, then float64 + renormalise
        e = np.exp(z, dtype=np.float32)
        probs_np = (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float64)
        row_sums = probs_np.sum(axis=1)
        max_drift = float(np.max(np.abs(row_sums - 1.0))) if n else 0.0
        if max_drift > _EXPECTED_FP_DRIFT_TOL:
            warnings.warn(f"predict_proba row sums deviate from 1.0 by up to {max_drift:.2e}, exceeding the expected float32 "
                          f"softmax drift bound ({_EXPECTED_FP_DRIFT_TOL:.0e}). Renormalizing anyway, but this likely indicates "
                          f"a numerical issue (extreme logits, NaN/Inf, or a bypassed softmax) rather than rounding.",
                          RuntimeWarning, stacklevel=2)
        probs_np /= row_sums[:, np.newaxis]
        return probs_np

    def predict_proba(self, X) -> np.ndarray:
        return self._forward_probs(X)

    def predict(self, X) -> np.ndarray:
        return self.classes_[np.argmax(self._forward_probs(X), axis=1)]

    # ---- parameters ----------------------------------------------------------------------------------------------
    def parameters(self):
        """-> (weights, biases): lists of float32 arrays in ``nn.Linear`` layout, read back from the device."""
        if not self._fitted():
            raise RuntimeError("TorchMLPClassifier is not fitted.")
        ws = [np.empty((o, i), np.float32) for i, o in zip(self._dims[:-1], self._dims[1:])]
        bs = [np.empty((o,), np.float32) for o in self._dims[1:]]
        _lib.check(_lib.lib().mmc_trainer_get_params(self._h, _ptr_array(ws), _ptr_array(bs)))
        return ws, bs

    @property
    def _module(self):
        """A torch module with the reference's ``_MLPModule`` layout (``.linears``), built from the current device
        parameters -- what ``build_calibrated_head`` (inference/head.py) and the export path read."""
        import torch
        import torch.nn as nn
        ws, bs = self.parameters()

        class _MLPModule(nn.Module):
            def __init__(self):
                super().__init__()
                self.linears = nn.ModuleList([nn.Linear(w.shape[1], w.shape[0]) for w in ws])
                with torch.no_grad():
                    for m, w, b in zip(self.linears, ws, bs):
                        m.weight.copy_(torch.from_numpy(w))
                        m.bias.copy_(torch.from_numpy(b))

            def forward(self, x):
                for i, m in enumerate(self.linears):
                    x = m(x)
                    if i < len(self.linears) - 1:
                        x = torch.relu(x)
                return x

        return _MLPModule().eval()

    def _adam_state(self):
        out = {}
        step = C.c_longlong(0)
        for which, name in ((0, "exp_avg"), (1, "exp_avg_sq")):
            ws = [np.empty((o, i), np.float32) for i, o in zip(self._dims[:-1], self._dims[1:])]
            bs = [np.empty((o,), np.float32) for o in self._dims[1:]]
            _lib.check(_lib.lib().mmc_trainer_adam_state(self._h, which, 0, _ptr_array(ws), _ptr_array(bs), C.byref(step)))
            out[name] = (ws, bs)
        out["step"] = int(step.value)
        return out

    def get_params(self, deep: bool = True) -> dict:
        return {"hidden_layer_sizes": self.hidden_layer_sizes, "activation": self.activation, "solver": self.solver,
                "alpha": self.alpha, "batch_size": self.batch_size, "learning_rate_init": self.learning_rate_init,
                "max_iter": self.max_iter, "shuffle": self.shuffle, "random_state": self.random_state, "tol": self.tol,
                "beta_1": self.beta_1, "beta_2": self.beta_2, "epsilon": self.epsilon,
                "class_weight": getattr(self, "class_weight", None)}

    def set_params(self, **params: Any) -> "TorchMLPClassifier":
        for key, value in params.items():
            if not hasattr(self, key):
                raise ValueError(f"Invalid parameter {key!r} for TorchMLPClassifier")
            setattr(self, key, value)
        return self

    # ---- pickling: parameters + optimizer state as arrays, rebuilt on load (torch_classifier.py:404-437) ------------
    def __getstate__(self) -> dict:
        state = {k: v for k, v in self.__dict__.items() if k != "_h"}
        if self._fitted():
            state["_module_state"] = self.parameters()
            state["_optimizer_state"] = self._adam_state()
        return state

    def __setstate__(self, state: dict) -> None:
        params = state.pop("_module_state", None)
        opt = state.pop("_optimizer_state", None)
        self.__dict__.update(state)
        self._h = None
        if params is not None:
            self._create_trainer(*params)
            if opt is not None:
                step = C.c_longlong(opt["step"])
                for which, name in ((0, "exp_avg"), (1, "exp_avg_sq")):
                    ws, bs = opt[name]
                    _lib.check(_lib.lib().mmc_trainer_adam_state(self._h, which, 1, _ptr_array(ws), _ptr_array(bs), C.byref(step)))

    def _release(self) -> None:
        if self._fitted():
            _lib.lib().mmc_trainer_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
