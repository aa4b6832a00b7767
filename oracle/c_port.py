"""ctypes wrapper of oracle/kge_oracle_c.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkge_oracle_c.so")
_MODELS = {"DistMult": 0, "TransE": 1, "ComplEx": 2}
_LOSSES = {"pointwize_hinge": 0, "pointwize_logistic": 1}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.kge_oracle_grads.restype = C.c_float
        _lib.kge_oracle_grads.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.kge_oracle_apply.restype = None
        _lib.kge_oracle_apply.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_int64]
        _lib.kge_oracle_threads.restype = C.c_int
        _lib.kge_oracle_set_threads.argtypes = [C.c_int]
        # a 1-GPU box gets a 16-core CPU share; wider teams only add private-table reduction work
        _lib.kge_oracle_set_threads(min(16, os.cpu_count() or 1, int(os.environ.get("KGE_ORACLE_THREADS", "16"))))
    return _lib


class CTrainer:
    """fp32 tables + optimizer state on the host; step() = one reference training batch."""

    def __init__(self, model, E, Rel, optimizer="adam", loss="pointwize_hinge", margin=1.0, norm=1, gamma=12.0):
        self.model, self.loss, self.margin, self.norm, self.gamma = _MODELS[model], _LOSSES[loss], float(margin), int(norm), float(gamma)
        self.E = np.ascontiguousarray(E, dtype=np.float32).copy()
        self.Rel = np.ascontiguousarray(Rel, dtype=np.float32).copy()
        self.gE, self.gR = np.zeros_like(self.E), np.zeros_like(self.Rel)
        self.kind = 1 if optimizer == "adam" else 0
        self.slots = [np.zeros_like(self.E), np.zeros_like(self.E), np.zeros_like(self.Rel), np.zeros_like(self.Rel)]
        self.lr, self.b1, self.b2, self.eps = (1e-3, 0.9, 0.999, 1e-7) if self.kind else (0.01, 0.0, 0.0, 0.0)
        self.t = 0

    def step(self, pos, neg):
        L = lib()
        pos = np.ascontiguousarray(pos, dtype=np.int32).reshape(-1, 3)
        neg = np.ascontiguousarray(neg, dtype=np.int32).reshape(-1, 3)
        N, d = self.E.shape
        loss = L.kge_oracle_grads(self.model, self.norm, self.gamma, self.loss, self.margin, self.E.ctypes.data, self.Rel.ctypes.data, N,
                                  self.Rel.shape[0], d, pos.ctypes.data, len(pos), neg.ctypes.data, len(neg), self.gE.ctypes.data,
                                  self.gR.ctypes.data, None)
        self.t += 1
        L.kge_oracle_apply(self.kind, self.E.ctypes.data, self.gE.ctypes.data, self.slots[0].ctypes.data, self.slots[1].ctypes.data,
                           self.E.size, self.lr, self.b1, self.b2, self.eps, self.t)
        L.kge_oracle_apply(self.kind, self.Rel.ctypes.data, self.gR.ctypes.data, self.slots[2].ctypes.data, self.slots[3].ctypes.data,
                           self.Rel.size, self.lr, self.b1, self.b2, self.eps, self.t)
        return float(loss)


def threads():
    return lib().kge_oracle_threads()
