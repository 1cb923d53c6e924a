#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container, where /root/reference is mounted.  Nothing in
tests/, bench.py or smoke() imports this file; the GPU box never sees the
reference.  Two kinds of fixture are produced:

1. Known answers the reference's own tests hold as inline literals
   (NeMo/tests/collections/asr/numba/rnnt_loss/test_rnnt_pytorch.py:85-127,
   :194-308, :362-401).  They are lifted out of the test file's AST (the file is
   read as text, never imported) and stored as JSON data.
2. Input/output vectors obtained by RUNNING the reference files that load
   standalone here: rnnt_numpy.py, multi_head_attention.py,
   causal_convs.py, common/parts/rnn.py.  Inputs are seeded; outputs are what the
   reference code returned in this container.

Usage:  python tests/golden/make_golden.py
"""
import ast
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
NEMO = os.path.join(REF, "NeMo")
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------- 1
def _literal(node):
    """Evaluate `np.array(<literal>)[.astype(..)]` / plain literal AST nodes."""
    if isinstance(node, ast.Call):
        f = node.func
        if isinstance(f, ast.Attribute) and f.attr == "astype":
            return _literal(f.value)
        if isinstance(f, ast.Attribute) and f.attr == "array":
            return _literal(node.args[0])
        raise ValueError(ast.dump(node)[:80])
    return ast.literal_eval(node)


def extract_known_answers():
    path = os.path.join(NEMO, "tests/collections/asr/numba/rnnt_loss/test_rnnt_pytorch.py")
    tree = ast.parse(open(path).read())
    want = {
        "test_case_small": ("acts", "labels", "expected_cost", "expected_grads"),
        "test_case_big_tensor": ("activations", "labels", "expected_costs", "expected_grads"),
        "test_case_small_clamp": ("acts", "labels", "expected_cost", "expected_grads", "GRAD_CLAMP"),
    }
    out = {}
    for cls in [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TestRNNTLossPytorch"]:
        for fn in [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want]:
            rec = {}
            for st in ast.walk(fn):
                if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
                    nm = st.targets[0].id
                    if nm in want[fn.name] and nm not in rec:
                        try:
                            rec[nm] = _literal(st.value)
                        except Exception:
                            pass
            rec["source"] = f"NeMo/tests/collections/asr/numba/rnnt_loss/test_rnnt_pytorch.py:{fn.lineno}"
            out[fn.name] = rec
    # normalise key names
    big = out["test_case_big_tensor"]
    big["acts"] = big.pop("activations")
    for k in out.values():
        k["blank"] = 0
    json.dump(out, open(os.path.join(HERE, "rnnt_known_answers.json"), "w"), indent=1)
    print("rnnt_known_answers.json:", {k: sorted(v) for k, v in out.items()})


# --------------------------------------------------------------------------- 2
def rnnt_numpy_cases():
    rn = _load(os.path.join(NEMO, "nemo/collections/asr/parts/numba/rnnt_loss/rnnt_numpy.py"), "ref_rnnt_numpy")
    cases = {}

    def run(name, acts, labels, flen, glen, blank, fastemit=0.0):
        acts_t = torch.tensor(acts, dtype=torch.float32)
        logp = torch.log_softmax(acts_t, -1).numpy()
        costs, grads_lp = rn.transduce_batch(logp, labels, flen, glen, blank, fastemit)
        B = acts.shape[0]
        alphas = np.zeros(acts.shape[:3], np.float32)
        betas = np.zeros(acts.shape[:3], np.float32)
        for b in range(B):
            t, u = int(flen[b]), int(glen[b]) + 1
            a, _ = rn.forward_pass(logp[b, :t, :u], labels[b, : u - 1], blank)
            bt, _ = rn.backward_pass(logp[b, :t, :u], labels[b, : u - 1], blank)
            alphas[b, :t, :u] = a
            betas[b, :t, :u] = bt
        # gradient w.r.t. the *logits* through the reference's own autograd path
        # (rnnt_numpy.RNNTLoss = log_softmax + _RNNT Function), cost = sum
        acts_g = torch.tensor(acts, dtype=torch.float32, requires_grad=True)
        loss = rn.RNNTLoss(blank=blank, fastemit_lambda=fastemit)(
            acts_g, torch.tensor(labels, dtype=torch.int64), torch.tensor(flen, dtype=torch.int64),
            torch.tensor(glen, dtype=torch.int64))
        loss.sum().backward()
        cases[name] = dict(acts=acts.astype(np.float32), labels=labels.astype(np.int64), flen=flen.astype(np.int64),
                           glen=glen.astype(np.int64), blank=np.int64(blank), fastemit=np.float32(fastemit),
                           costs=np.asarray(costs, np.float32), grads_logprobs=np.asarray(grads_lp, np.float32),
                           grads_logits=acts_g.grad.numpy().astype(np.float32), alphas=alphas, betas=betas)

    # recipes of the reference tests (RandomState(0)) -- test_rnnt_pytorch.py:146-148, :327-334
    rng = np.random.RandomState(0)
    run("small_random_1x4x3x3", rng.randn(1, 4, 3, 3), np.array([[1, 2]]), np.array([4]), np.array([2]), 0)
    rng = np.random.RandomState(0)
    run("large_random_4x8x11x5", rng.randn(4, 8, 11, 5),
        np.array([[1, 2, 4, 3, 2, 2, 1, 1, 1, 1], [3, 2, 2, 3, 4, 1, 1, 1, 1, 1],
                  [4, 4, 1, 2, 1, 3, 4, 3, 1, 2], [1, 1, 2, 1, 2, 3, 3, 1, 1, 1]]),
        np.array([8] * 4), np.array([10] * 4), 0)
    # test_gpu_rnnt_kernel.py:56-75 recipe: 1x5x11x3, labels all ones
    rng = np.random.RandomState(0)
    run("kernel_1x5x11x3", rng.randn(1, 5, 11, 3), np.ones((1, 10), np.int64), np.array([5]), np.array([10]), 0)
    # fastemit variant (test_rnnt_pytorch.py:168-184)
    rng = np.random.RandomState(0)
    run("small_random_fastemit_0.01", rng.randn(1, 4, 3, 3), np.array([[1, 2]]), np.array([4]), np.array([2]), 0, 0.01)
    # ragged batch, blank = last index, V+1 = 257 as in the model (hybrid_rnnt_ctc_bpe_models.py:118-124)
    rng = np.random.RandomState(1234)
    B, T, U, V1 = 3, 13, 7, 257
    flen = np.array([13, 9, 5]); glen = np.array([4, 6, 1])
    run("ragged_3x13x7x257_blank256", rng.randn(B, T, U, V1) * 2.0, rng.randint(0, 256, size=(B, U - 1)),
        flen, glen, 256)
    # U = 1 (empty transcript) and T = 1 edge cases
    rng = np.random.RandomState(7)
    run("edge_empty_label", rng.randn(2, 6, 1, 9), np.zeros((2, 0), np.int64), np.array([6, 3]), np.array([0, 0]), 8)
    rng = np.random.RandomState(8)
    run("edge_T1", rng.randn(2, 1, 4, 9), rng.randint(0, 8, size=(2, 3)), np.array([1, 1]), np.array([3, 2]), 8)
    flat = {}
    for k, v in cases.items():
        for kk, vv in v.items():
            flat[f"{k}/{kk}"] = vv
    np.savez_compressed(os.path.join(HERE, "rnnt_numpy_cases.npz"), **flat)
    print("rnnt_numpy_cases.npz:", list(cases))


def _asr_pkg_stubs():
    """Empty package objects so reference files with package-path imports of each
    other (multi_head_attention -> nemo.utils, causal_convs) load by file path."""
    if NEMO not in sys.path:
        sys.path.insert(0, NEMO)


def module_cases():
    _asr_pkg_stubs()
    sub = os.path.join(NEMO, "nemo/collections/asr/parts/submodules")
    mha = _load(os.path.join(sub, "multi_head_attention.py"), "ref_mha")
    cc = _load(os.path.join(sub, "causal_convs.py"), "ref_causal_convs")
    rnn = _load(os.path.join(NEMO, "nemo/collections/common/parts/rnn.py"), "ref_rnn")
    out = {}
    torch.manual_seed(1234)
    # --- RelPositionalEncoding + RelPositionMultiHeadAttention (multi_head_attention.py:157-250, 935-979)
    B, T, d, h = 3, 11, 32, 4
    pos = mha.RelPositionalEncoding(d_model=d, dropout_rate=0.0, max_len=64, xscale=d ** 0.5, dropout_rate_emb=0.0)
    pos.extend_pe(64, torch.device("cpu"))
    att = mha.RelPositionMultiHeadAttention(n_head=h, n_feat=d, dropout_rate=0.0, pos_bias_u=None, pos_bias_v=None)
    with torch.no_grad():
        att.pos_bias_u.normal_(0, 0.3)
        att.pos_bias_v.normal_(0, 0.3)
    att.eval(); pos.eval()
    x = torch.randn(B, T, d)
    lens = torch.tensor([11, 7, 4])
    xs, pos_emb = pos(x)
    valid = torch.arange(T)[None, :] < lens[:, None]
    att_mask = ~(valid[:, :, None] & valid[:, None, :])
    xs_g = xs.clone().requires_grad_(True)
    y = att(xs_g, xs_g, xs_g, att_mask, pos_emb)
    gy = torch.randn_like(y)
    y.backward(gy)
    out.update({"mha/x": x, "mha/lens": lens, "mha/xscaled": xs, "mha/pos_emb": pos_emb, "mha/y": y.detach(),
                "mha/gy": gy, "mha/gx": xs_g.grad})
    for n, p in att.named_parameters():
        out[f"mha/param/{n}"] = p.detach()
        out[f"mha/grad/{n}"] = p.grad
    # --- CausalConv1D depthwise k=31 pad 15/15 (causal_convs.py:72-150; conformer_modules.py:311-319)
    C, k = 16, 31
    conv = cc.CausalConv1D(C, C, kernel_size=k, stride=1, padding=(k - 1) // 2, groups=C, bias=True)
    xc = torch.randn(2, C, 40)
    out.update({"dwconv/x": xc, "dwconv/w": conv.weight.detach(), "dwconv/b": conv.bias.detach(),
                "dwconv/y": conv(xc).detach()})
    # --- label_collate + LSTMDropout prediction RNN (common/parts/rnn.py:151-235, 536-561)
    H = 24
    lstm = rnn.rnn(input_size=H, hidden_size=H, num_layers=1, norm=None, forget_gate_bias=1.0, dropout=0.0)
    lstm.eval()
    xin = torch.randn(9, 3, H)  # (U+1, B, H)
    g, (hn, cn) = lstm(xin, None)
    out.update({"lstm/x": xin, "lstm/y": g.detach(), "lstm/h": hn.detach(), "lstm/c": cn.detach()})
    for n, p in lstm.named_parameters():
        out[f"lstm/param/{n}"] = p.detach()
    coll = rnn.label_collate([[1, 2, 3], [4], [5, 6]])
    out["label_collate/out"] = coll
    np.savez_compressed(os.path.join(HERE, "module_cases.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("module_cases.npz:", len(out), "arrays")


def _empty_packages(*names):
    """Empty package objects (only `__path__`, so that sub-module FILES still resolve) for the packages whose
    `__init__.py` pulls in hydra / lightning / numba: lets subsampling.py -- whose only blocker is the package-path
    import of causal_convs -- load unmodified ('optional technique')."""
    for name in names:
        if name in sys.modules:
            continue
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(NEMO, *name.split("."))]
        sys.modules[name] = mod


def subsampling_cases():
    """ConvSubsampling ('striding', x4) + calc_length (A/parts/submodules/subsampling.py:217-253,385-437,566-576) RUN
    here on seeded inputs: pins SURVEY row a6 and the frame-count rule on the reference's own arithmetic."""
    _asr_pkg_stubs()
    import nemo.utils  # noqa: F401  (imports fine here: SURVEY 8(c))
    _empty_packages("nemo.collections", "nemo.collections.asr", "nemo.collections.asr.parts",
                    "nemo.collections.asr.parts.submodules")
    sub = _load(os.path.join(NEMO, "nemo/collections/asr/parts/submodules/subsampling.py"), "ref_subsampling")
    out = {}
    torch.manual_seed(4321)
    for tag, (feat_in, C, d, B, Tm) in {"a": (80, 16, 24, 3, 61), "b": (80, 8, 16, 2, 37)}.items():
        m = sub.ConvSubsampling(subsampling="striding", subsampling_factor=4, feat_in=feat_in, feat_out=d, conv_channels=C,
                                subsampling_conv_chunking_factor=1, activation=torch.nn.ReLU(True), is_causal=False)
        m.eval()
        x = torch.randn(B, Tm, feat_in)
        lens = torch.tensor([Tm, max(1, Tm - 9), max(1, Tm // 2)][:B])
        y, ylen = m(x, lens)
        out[f"sub/{tag}/x"], out[f"sub/{tag}/lens"] = x, lens
        out[f"sub/{tag}/y"], out[f"sub/{tag}/ylen"] = y.detach(), ylen
        for n, p in m.named_parameters():
            out[f"sub/{tag}/param/{n}"] = p.detach()
    n = torch.arange(1, 3100)
    out["calc_length/n"] = n
    out["calc_length/out"] = sub.calc_length(n, all_paddings=2, kernel_size=3, stride=2, ceil_mode=False, repeat_num=2)
    np.savez_compressed(os.path.join(HERE, "subsampling_cases.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("subsampling_cases.npz:", len(out), "arrays")


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted; fixtures are generated in the build container only"
    extract_known_answers()
    rnnt_numpy_cases()
    module_cases()
    subsampling_cases()
