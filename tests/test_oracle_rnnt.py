"""Pins oracle/rnnt_ref.c against the reference's own known answers and rnnt_numpy outputs (CPU only)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import rnnt_oracle as orc

from conftest import GOLDEN


def _cases():
    z = np.load(os.path.join(GOLDEN, "rnnt_numpy_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


KA = json.load(open(os.path.join(GOLDEN, "rnnt_known_answers.json")))


def test_known_answer_small():
    k = KA["test_case_small"]
    acts = np.array(k["acts"], np.float32)
    r = orc.rnnt_loss(acts, np.array(k["labels"]), [acts.shape[1]], [2], blank=0)
    assert np.allclose(r["costs"].sum(), k["expected_cost"], atol=1e-6, rtol=1e-6)
    assert np.allclose(r["grads"], np.array(k["expected_grads"]), atol=1e-6, rtol=1e-5)


def test_known_answer_big_tensor():
    k = KA["test_case_big_tensor"]
    acts = np.array(k["acts"], np.float32)
    B, T, U1, V = acts.shape
    r = orc.rnnt_loss(acts, np.array(k["labels"]), [T] * B, [U1 - 1] * B, blank=0)
    assert np.allclose(r["costs"], k["expected_costs"], atol=1e-5)
    assert np.allclose(r["grads"], np.array(k["expected_grads"]), atol=1e-6, rtol=1e-3)


def test_known_answer_clamp():
    k = KA["test_case_small_clamp"]
    acts = np.array(k["acts"], np.float32)
    r = orc.rnnt_loss(acts, np.array(k["labels"]), [acts.shape[1]], [2], blank=0, clamp=k["GRAD_CLAMP"])
    assert np.allclose(r["costs"].sum(), k["expected_cost"], atol=1e-6, rtol=1e-5)
    assert np.allclose(r["grads"], np.array(k["expected_grads"]), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", _cases()[1])
def test_vs_reference_rnnt_numpy(name):
    z, _ = _cases()
    g = lambda k: z[f"{name}/{k}"]
    r = orc.rnnt_loss(g("acts"), g("labels"), g("flen"), g("glen"), int(g("blank")), float(g("fastemit")),
                      want_lp_grads=True)
    assert np.allclose(r["costs"], g("costs"), atol=2e-5, rtol=1e-5)
    assert np.allclose(r["alphas"], g("alphas"), atol=2e-5, rtol=1e-5)
    assert np.allclose(r["betas"], g("betas"), atol=2e-5, rtol=1e-5)
    assert np.allclose(r["grads_lp"], g("grads_logprobs"), atol=1e-5, rtol=1e-4)
    if float(g("fastemit")) == 0.0:
        # with FastEmit the GPU kernel adds an extra term the numpy/autograd path does not (gpu_rnnt_kernel.py:363-375)
        assert np.allclose(r["grads"], g("grads_logits"), atol=1e-5, rtol=1e-4)


def test_c_vs_pure_numpy_and_autograd():
    rng = np.random.RandomState(5)
    B, T, U1, V = 2, 6, 4, 7
    acts = rng.randn(B, T, U1, V).astype(np.float32)
    labels = rng.randint(0, V - 1, size=(B, U1 - 1))
    flen, glen = np.array([6, 4]), np.array([3, 2])
    r = orc.rnnt_loss(acts, labels, flen, glen, blank=V - 1, want_lp_grads=True)
    c2, g2 = orc.rnnt_loss_numpy(acts, labels, flen, glen, blank=V - 1)
    assert np.allclose(r["costs"], c2, atol=1e-5)
    assert np.allclose(r["grads"], g2, atol=1e-5)
    # log-prob grads pushed through log_softmax by autograd == fused logits grads (cpu path == gpu path)
    x = torch.tensor(acts, requires_grad=True)
    torch.log_softmax(x, -1).backward(torch.tensor(r["grads_lp"]))
    assert np.allclose(x.grad.numpy(), r["grads"], atol=1e-5)


def test_ctc_vs_torch():
    torch.manual_seed(3)
    T, B, V, S = 12, 3, 9, 5
    lp = torch.randn(T, B, V).log_softmax(-1).requires_grad_(True)
    tg = torch.randint(0, V - 1, (B, S))
    tg[1, 1] = tg[1, 0]  # repeated label
    il = torch.tensor([12, 9, 3]); tl = torch.tensor([5, 3, 4])  # last one infeasible -> inf -> zeroed
    loss = torch.nn.functional.ctc_loss(lp, tg, il, tl, blank=V - 1, reduction="none", zero_infinity=True)
    loss.sum().backward()
    nll, grad = orc.ctc_loss(lp.detach().numpy(), tg.numpy(), il.numpy(), tl.numpy(), V - 1)
    assert np.allclose(nll, loss.detach().numpy(), atol=1e-5)
    assert np.allclose(grad, lp.grad.numpy(), atol=1e-5)
