"""ctypes front end to oracle/rnnt_ref.c plus a pure-numpy restatement for tiny cases.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity pinned by tests/test_oracle_rnnt.py against
the reference's known answers (tests/golden/rnnt_known_answers.json) and rnnt_numpy outputs
(tests/golden/rnnt_numpy_cases.npz).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        _LIB.oracle_rnnt_loss.argtypes = [f32p, i64p, i64p, i64p] + [ctypes.c_int] * 5 + [ctypes.c_float] * 2 + [f32p] * 5
        _LIB.oracle_rnnt_loss.restype = ctypes.c_int
        _LIB.oracle_ctc_loss.argtypes = [f32p, i64p, i64p, i64p] + [ctypes.c_int] * 6 + [f32p, f32p]
        _LIB.oracle_ctc_loss.restype = ctypes.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def rnnt_loss(logits, labels, flen, glen, blank, fastemit=0.0, clamp=0.0, want_lp_grads=False):
    """Returns dict(costs[B], grads[B,T,U1,V] wrt logits, alphas, betas[, grads_lp])."""
    logits = np.ascontiguousarray(logits, np.float32)
    B, T, U1, V = logits.shape
    labels = np.ascontiguousarray(labels, np.int64).reshape(B, U1 - 1)
    flen = np.ascontiguousarray(flen, np.int64)
    glen = np.ascontiguousarray(glen, np.int64)
    costs = np.zeros(B, np.float32)
    grads = np.zeros_like(logits)
    alphas = np.zeros((B, T, U1), np.float32)
    betas = np.zeros((B, T, U1), np.float32)
    glp = np.zeros_like(logits) if want_lp_grads else None
    rc = lib().oracle_rnnt_loss(_p(logits, ctypes.c_float), _p(labels, ctypes.c_int64), _p(flen, ctypes.c_int64),
                                _p(glen, ctypes.c_int64), B, T, U1, V, int(blank), float(fastemit), float(clamp),
                                _p(costs, ctypes.c_float), _p(grads, ctypes.c_float), _p(alphas, ctypes.c_float),
                                _p(betas, ctypes.c_float), _p(glp, ctypes.c_float))
    if rc != 0:
        raise ValueError("oracle_rnnt_loss: invalid sizes")
    out = dict(costs=costs, grads=grads, alphas=alphas, betas=betas)
    if want_lp_grads:
        out["grads_lp"] = glp
    return out


def ctc_loss(log_probs_tbv, targets, in_len, tg_len, blank, zero_infinity=True):
    lp = np.ascontiguousarray(log_probs_tbv, np.float32)
    T, B, V = lp.shape
    targets = np.ascontiguousarray(targets, np.int64)
    S = targets.shape[1] if targets.ndim == 2 else 0
    in_len = np.ascontiguousarray(in_len, np.int64)
    tg_len = np.ascontiguousarray(tg_len, np.int64)
    nll = np.zeros(B, np.float32)
    grad = np.zeros_like(lp)
    rc = lib().oracle_ctc_loss(_p(lp, ctypes.c_float), _p(targets, ctypes.c_int64), _p(in_len, ctypes.c_int64),
                               _p(tg_len, ctypes.c_int64), T, B, V, S, int(blank), int(zero_infinity),
                               _p(nll, ctypes.c_float), _p(grad, ctypes.c_float))
    if rc != 0:
        raise ValueError("oracle_ctc_loss: invalid sizes")
    return nll, grad


# --------------------------------------------------------------------------------------------
# pure-numpy restatement (rnnt_numpy.py:112-207 semantics) -- second, independent check of the C
def rnnt_loss_numpy(logits, labels, flen, glen, blank):
    logits = np.asarray(logits, np.float64)
    B, T, U1, V = logits.shape
    m = logits.max(-1, keepdims=True)
    lp = logits - m - np.log(np.exp(logits - m).sum(-1, keepdims=True))
    costs = np.zeros(B)
    grads = np.zeros_like(lp)
    for b in range(B):
        Tb, Ub = int(flen[b]), int(glen[b]) + 1
        a = np.zeros((Tb, Ub)); be = np.zeros((Tb, Ub))
        for t in range(Tb):
            for u in range(Ub):
                if t == 0 and u == 0:
                    continue
                c = []
                if t > 0:
                    c.append(a[t - 1, u] + lp[b, t - 1, u, blank])
                if u > 0:
                    c.append(a[t, u - 1] + lp[b, t, u - 1, labels[b][u - 1]])
                a[t, u] = np.logaddexp.reduce(c)
        ll = a[Tb - 1, Ub - 1] + lp[b, Tb - 1, Ub - 1, blank]
        for t in reversed(range(Tb)):
            for u in reversed(range(Ub)):
                if t == Tb - 1 and u == Ub - 1:
                    be[t, u] = lp[b, t, u, blank]
                    continue
                c = []
                if t < Tb - 1:
                    c.append(be[t + 1, u] + lp[b, t, u, blank])
                if u < Ub - 1:
                    c.append(be[t, u + 1] + lp[b, t, u, labels[b][u]])
                be[t, u] = np.logaddexp.reduce(c)
        costs[b] = -ll
        for t in range(Tb):
            for u in range(Ub):
                g = np.exp(a[t, u] + be[t, u] + lp[b, t, u] - ll)
                if t < Tb - 1:
                    g[blank] -= np.exp(a[t, u] + lp[b, t, u, blank] + be[t + 1, u] - ll)
                else:
                    if u == Ub - 1:
                        g[blank] -= np.exp(a[t, u] + lp[b, t, u, blank] - ll)
                if u < Ub - 1:
                    l = labels[b][u]
                    g[l] -= np.exp(a[t, u] + lp[b, t, u, l] + be[t, u + 1] - ll)
                grads[b, t, u] = g
    return costs, grads
