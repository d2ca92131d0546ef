"""Serve-time loader for the portable classifier artifact (model.pt + model.json), with the
calibrated MLP head running on the MI355X.

Mirrors reference ``mermaid_classifier/pyspacer/inference/loader.py:16-75``
(``Predictor`` / ``load_predictor``) and the constants of ``inference/__init__.py:9-32``:
same manifest validation and exceptions (``ManifestError`` on schema-version, class-count or
input_dim mismatch; ``ValueError`` on a wrongly shaped batch), same return type
(``predict_proba`` -> float64 ``(N, K)`` of an fp32 computation), ``classes`` / ``classes_`` /
``input_dim`` attributes.

``model.pt`` is the frozen TorchScript ``CalibratedHead`` (inference/export.py:54-57,90-91).
Freezing inlines parameters as graph constants, so the Linear weights/biases and the Platt
vectors ``a``/``b`` are read from the graph's ``aten::linear`` / ``aten::mul`` / ``aten::add``
nodes, and the HIP head is then checked against the TorchScript graph itself on a probe batch
at load time -- a graph this loader cannot reproduce is refused loudly, never served.
"""

from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Any, List, Sequence

import numpy as np

from . import _lib
from .backbone import _current_stream_ptr, _device_index

SCHEMA_VERSION = 1
TASK_NAME = "pyspacer_mlp_classifier"


class ManifestError(Exception):
    """model.json is incompatible with the graph (schema version, class count, input_dim),
    or the graph is not a CalibratedHead this loader can reproduce."""


class HeadParams:
    def __init__(self, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], a: np.ndarray, b: np.ndarray):
        self.weights = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self.biases = [np.ascontiguousarray(v, dtype=np.float32) for v in biases]
        self.a = np.ascontiguousarray(a, dtype=np.float32)
        self.b = np.ascontiguousarray(b, dtype=np.float32)
        if len(self.weights) != len(self.biases) or not self.weights:
            raise ValueError("weights and biases must be non-empty and of equal length")
        if self.a.ndim != 1 or self.a.shape != self.b.shape:
            raise ValueError(f"a and b must be 1-D and of equal shape; got {self.a.shape}, {self.b.shape}")
        for i, (w, v) in enumerate(zip(self.weights, self.biases)):
            if w.ndim != 2 or v.shape != (w.shape[0],):
                raise ValueError(f"layer {i}: weight {w.shape} / bias {v.shape} mismatch")
            if i and w.shape[1] != self.weights[i - 1].shape[0]:
                raise ValueError(f"layer {i}: in_features {w.shape[1]} != previous out_features")
        if self.weights[-1].shape[0] != self.a.shape[0]:
            raise ValueError("last layer width != number of calibrators")

    @property
    def dims(self) -> List[int]:
        return [self.weights[0].shape[1]] + [w.shape[0] for w in self.weights]


def params_from_torchscript(graph) -> HeadParams:
    """Read Linear weights/biases and Platt a/b out of a frozen scripted CalibratedHead."""
    g = graph.graph
    weights, biases, a, b = [], [], None, None
    softmax_out = None
    mul_out = None
    for node in g.nodes():
        kind = node.kind()
        if kind == "aten::linear":
            ins = list(node.inputs())
            w, bias = ins[1].toIValue(), ins[2].toIValue()
            if w is None or bias is None:
                raise ManifestError("aten::linear with non-constant parameters: graph is not frozen")
            weights.append(w.detach().cpu().numpy())
            biases.append(bias.detach().cpu().numpy())
        elif kind == "aten::softmax":
            softmax_out = node.output().debugName()
        elif kind == "aten::mul" and softmax_out is not None and a is None:
            ins = list(node.inputs())
            names = [i.debugName() for i in ins]
            if softmax_out in names:
                other = ins[1 - names.index(softmax_out)].toIValue()
                if other is not None:
                    a = other.detach().cpu().numpy()
                    mul_out = node.output().debugName()
        elif kind == "aten::add" and mul_out is not None and b is None:
            ins = list(node.inputs())
            names = [i.debugName() for i in ins[:2]]
            if mul_out in names:
                other = ins[1 - names.index(mul_out)].toIValue()
                if other is not None:
                    b = other.detach().cpu().numpy()
    if not weights or a is None or b is None:
        raise ManifestError("model.pt is not a frozen CalibratedHead graph (linear/softmax/mul/add pattern not found)")
    try:
        return HeadParams(weights, biases, a, b)
    except ValueError as exc:
        raise ManifestError(f"inconsistent parameters in model.pt: {exc}") from exc


class DeviceHead:
    """mmc_head_* handle wrapper."""

    def __init__(self, params: HeadParams, device="cuda"):
        lib = _lib.lib()
        self.params = params
        n = len(params.weights)
        fp = C.POINTER(C.c_float)
        W = (fp * n)(*[w.ctypes.data_as(fp) for w in params.weights])
        B = (fp * n)(*[v.ctypes.data_as(fp) for v in params.biases])
        dims = (C.c_int * (n + 1))(*params.dims)
        self.device_index = _device_index(device)
        self._h = C.c_void_p()
        _lib.check(lib.mmc_head_create(W, B, dims, n, params.a.ctypes.data_as(fp), params.b.ctypes.data_as(fp),
                                       int(params.a.shape[0]), self.device_index, C.byref(self._h)))
        self.input_dim = params.dims[0]
        self.n_classes = int(params.a.shape[0])

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().mmc_head_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def predict(self, feats, want_argmax: bool = True):
        """feats: (N,input_dim) float32, numpy (host) or torch cuda tensor.
        Returns (proba (N,K) float32, argmax (N,) int32) of the same kind."""
        lib = _lib.lib()
        st = _current_stream_ptr(self.device_index)
        if isinstance(feats, np.ndarray):
            x = np.ascontiguousarray(feats, dtype=np.float32)
            n = x.shape[0]
            proba = np.empty((n, self.n_classes), dtype=np.float32)
            arg = np.empty((n,), dtype=np.int32)
            if n:
                _lib.check(lib.mmc_head_predict(self._h, x.ctypes.data, n, proba.ctypes.data,
                                                arg.ctypes.data if want_argmax else None,
                                                _lib.MMC_IN_HOST | _lib.MMC_OUT_HOST, st))
            return proba, arg
        import torch
        x = feats.contiguous()
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.input_dim:
            raise ValueError(f"features must be a float32 cuda tensor (N, {self.input_dim}); got {tuple(x.shape)}")
        n = x.shape[0]
        proba = torch.empty((n, self.n_classes), dtype=torch.float32, device=x.device)
        arg = torch.empty((n,), dtype=torch.int32, device=x.device)
        if n:
            _lib.check(lib.mmc_head_predict(self._h, x.data_ptr(), n, proba.data_ptr(),
                                            arg.data_ptr() if want_argmax else None, 0, st))
        return proba, arg


class Predictor:
    """A loaded classifier head: feature batch -> calibrated probabilities (loader.py:16-35)."""

    def __init__(self, head: DeviceHead, classes: List[str], input_dim: int) -> None:
        self._head = head
        self.classes = classes
        self.input_dim = input_dim

    @property
    def classes_(self) -> List[str]:
        return self.classes

    def predict_proba(self, features: Any) -> np.ndarray:
        arr = np.asarray(features, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[1] != self.input_dim:
            raise ValueError(f"features must be (N, {self.input_dim}); got {arr.shape}.")
        proba, _ = self._head.predict(arr, want_argmax=False)
        return proba.astype(np.float64)

    def predict(self, features: Any) -> List[str]:
        arr = np.asarray(features, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[1] != self.input_dim:
            raise ValueError(f"features must be (N, {self.input_dim}); got {arr.shape}.")
        _, arg = self._head.predict(arr, want_argmax=True)
        return [self.classes[i] for i in arg.tolist()]


def load_predictor(model_pt_path, model_json_path, device="cuda", probe_tol: float = 1e-5) -> Predictor:
    """Load model.pt + model.json, validating compatibility loudly (loader.py:38-75)."""
    import torch

    manifest = json.loads(Path(model_json_path).read_text())
    schema_version = manifest.get("schema_version")
    if schema_version != SCHEMA_VERSION:
        raise ManifestError(
            f"model.json schema_version={schema_version!r} is incompatible"
            f" with this loader (expects {SCHEMA_VERSION}).")
    classes = manifest["classes"]
    input_dim = int(manifest["input_dim"])
    graph = torch.jit.load(str(model_pt_path), map_location="cpu")
    graph.eval()
    params = params_from_torchscript(graph)
    if params.dims[0] != input_dim:
        raise ManifestError(
            f"graph rejects input_dim={input_dim} declared in model.json: first Linear expects {params.dims[0]}")
    if params.a.shape[0] != len(classes):
        raise ManifestError(
            f"class-count mismatch: graph outputs {params.a.shape[0]} classes"
            f" but model.json declares {len(classes)}.")
    try:
        head = DeviceHead(params, device=device)
    except ValueError as exc:
        raise ManifestError(f"graph cannot be served: {exc}") from exc
    # probe: zeros (the reference's probe) plus a few seeded rows, HIP head vs the graph itself
    rng = np.random.default_rng(0)
    probe = np.concatenate([np.zeros((1, input_dim), np.float32),
                            rng.normal(0.3, 0.6, size=(7, input_dim)).astype(np.float32)])
    with torch.no_grad():
        want = graph(torch.from_numpy(probe)).numpy()
    got, _ = head.predict(probe, want_argmax=False)
    diff = float(np.max(np.abs(got - want)))
    if not np.isfinite(diff) or diff > probe_tol:
        raise ManifestError(f"HIP head diverges from model.pt on the load-time probe: max|dp|={diff:.3e} > {probe_tol:.1e}")
    return Predictor(head, list(classes), input_dim)
