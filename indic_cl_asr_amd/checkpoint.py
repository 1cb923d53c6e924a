"""Checkpoint interchange.

* trainable-only state dict `.pth` -- R/utils.py:265-271 (`save_model`) and its `load_state_dict(strict=False)` use in the
  CL scripts: `save_trainable` / `load_weights`.
* `.nemo` archives -- a tar (optionally gzipped) holding `model_config.yaml` and `model_weights.ckpt`
  (NeMo core/connectors/save_restore_connector.py): `read_nemo` returns (config dict, state dict) without importing
  NeMo; `model_from_nemo` builds the MI355X model from the encoder / prediction / joint / preprocessor sections of that
  yaml and loads the weights (parameter names are the reference's, so no key translation is needed).  `write_nemo` emits
  the same two members so that weights trained here load back into the reference.
* continual-learning state -- Fisher / omega / theta* flat buffers with their tensor table (`save_cl_state` /
  `load_cl_state`); the reference keeps these in process memory only and loses them on restart.
"""
import io
import os
import tarfile
from typing import Dict, Optional, Tuple

import torch
import yaml

from .config import ModelConfig, model_config


def save_trainable(model, path):
    """R/utils.py:265-271: only parameters with requires_grad, under their reference names."""
    m = getattr(model, "module", model)
    from . import cl
    cl.flush_pending_updates()
    torch.save({n: p.detach().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}, path)


def load_weights(model, path_or_state, strict=False):
    """Full or trainable-only state dict (file or dict) into the model; returns torch's (missing, unexpected) report."""
    from . import cl
    from .ops import fast
    cl.flush_pending_updates()   # a deferred data-parallel AdamW update must land BEFORE the new weights, not on top of them
    state = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, os.PathLike)) else path_or_state
    if isinstance(state, dict) and "state_dict" in state and not any(torch.is_tensor(v) for v in state.values()):
        state = state["state_dict"]          # Lightning-style wrapper
    report = getattr(model, "module", model).load_state_dict(state, strict=strict)
    fast.invalidate_weight_caches()          # bf16 weight images of the old values must not be served again
    return report


def read_nemo(path) -> Tuple[dict, Dict[str, torch.Tensor]]:
    cfg, state = None, None
    with tarfile.open(path, "r:*") as tar:
        for member in tar.getmembers():
            name = os.path.basename(member.name)
            if name == "model_config.yaml":
                cfg = yaml.safe_load(tar.extractfile(member).read())
            elif name == "model_weights.ckpt":
                state = torch.load(io.BytesIO(tar.extractfile(member).read()), map_location="cpu")
    if cfg is None or state is None:
        raise ValueError(f"{path}: not a .nemo archive (model_config.yaml + model_weights.ckpt expected)")
    return cfg, state


def config_from_nemo_yaml(cfg: dict, **overrides) -> ModelConfig:
    """The fields of NeMo's hybrid Conformer yaml that this path uses (conformer_hybrid_transducer_ctc_bpe.yaml:62-211)."""
    enc, pre = cfg.get("encoder", {}), cfg.get("preprocessor", {})
    pred = cfg.get("decoder", {}).get("prednet", {})
    jnt = cfg.get("joint", {}).get("jointnet", {})
    sa = cfg.get("spec_augment", {})
    kw = {}

    def put(dst, src, key, cast=lambda v: v):
        if key in src and src[key] is not None:
            kw[dst] = cast(src[key])

    put("feat_in", enc, "feat_in", int); put("d_model", enc, "d_model", int); put("n_layers", enc, "n_layers", int)
    put("n_heads", enc, "n_heads", int); put("ff_expansion_factor", enc, "ff_expansion_factor", int)
    put("conv_kernel_size", enc, "conv_kernel_size", int); put("pos_emb_max_len", enc, "pos_emb_max_len", int)
    put("dropout", enc, "dropout", float); put("dropout_pre_encoder", enc, "dropout_pre_encoder", float)
    put("dropout_emb", enc, "dropout_emb", float); put("dropout_att", enc, "dropout_att", float)
    put("pred_hidden", pred, "pred_hidden", int); put("pred_dropout", pred, "dropout", float)
    put("joint_hidden", jnt, "joint_hidden", int); put("joint_dropout", jnt, "dropout", float)
    put("fused_batch_size", cfg.get("joint", {}), "fused_batch_size", int)
    put("sample_rate", pre, "sample_rate", int); put("n_fft", pre, "n_fft", int); put("dither", pre, "dither", float)
    if "window_size" in pre and "sample_rate" in pre:
        kw["n_window_size"] = int(round(float(pre["window_size"]) * int(pre["sample_rate"])))
        kw["n_window_stride"] = int(round(float(pre.get("window_stride", 0.01)) * int(pre["sample_rate"])))
    put("freq_masks", sa, "freq_masks", int); put("time_masks", sa, "time_masks", int)
    put("freq_width", sa, "freq_width", int); put("time_width", sa, "time_width", float)
    put("ctc_loss_weight", cfg.get("aux_ctc", {}), "ctc_loss_weight", float)
    kw.update(overrides)
    return model_config(**kw)


def model_from_nemo(path, strict=False, **overrides):
    """(model, load report).  `overrides`: ModelConfig fields the archive's yaml does not carry (languages, compute_dtype …)."""
    from .model import EncDecHybridRNNTCTCModel
    cfg, state = read_nemo(path)
    model = EncDecHybridRNNTCTCModel(config_from_nemo_yaml(cfg, **overrides))
    return model, model.load_state_dict(state, strict=strict)


def nemo_yaml_from_config(c: ModelConfig) -> dict:
    return {
        "sample_rate": c.sample_rate,
        "preprocessor": {"sample_rate": c.sample_rate, "window_size": c.n_window_size / c.sample_rate,
                         "window_stride": c.n_window_stride / c.sample_rate, "features": c.feat_in, "n_fft": c.n_fft,
                         "dither": c.dither, "normalize": "per_feature", "window": "hann"},
        "spec_augment": {"freq_masks": c.freq_masks, "time_masks": c.time_masks, "freq_width": c.freq_width,
                         "time_width": c.time_width},
        "encoder": {"feat_in": c.feat_in, "n_layers": c.n_layers, "d_model": c.d_model, "subsampling": "striding",
                    "subsampling_factor": 4, "subsampling_conv_channels": c.d_model, "ff_expansion_factor": c.ff_expansion_factor,
                    "self_attention_model": "rel_pos", "n_heads": c.n_heads, "conv_kernel_size": c.conv_kernel_size,
                    "conv_norm_type": "batch_norm", "pos_emb_max_len": c.pos_emb_max_len, "dropout": c.dropout,
                    "dropout_pre_encoder": c.dropout_pre_encoder, "dropout_emb": c.dropout_emb, "dropout_att": c.dropout_att},
        "decoder": {"prednet": {"pred_hidden": c.pred_hidden, "pred_rnn_layers": 1, "dropout": c.pred_dropout}},
        "joint": {"fused_batch_size": c.fused_batch_size,
                  "jointnet": {"joint_hidden": c.joint_hidden, "activation": "relu", "dropout": c.joint_dropout}},
        "aux_ctc": {"ctc_loss_weight": c.ctc_loss_weight},
    }


def write_nemo(model, path):
    m = getattr(model, "module", model)
    from . import cl
    cl.flush_pending_updates()
    with tarfile.open(path, "w:gz") as tar:
        def add(name, data: bytes):
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
        add("model_config.yaml", yaml.safe_dump(nemo_yaml_from_config(m.cfg)).encode())
        buf = io.BytesIO()
        torch.save({k: v.detach().cpu() for k, v in m.state_dict().items()}, buf)
        add("model_weights.ckpt", buf.getvalue())


def save_cl_state(path, **flat_dicts):
    """Persist FlatDict buffers (Fisher, omega, theta*) with their tensor table; `None` values are skipped."""
    out = {}
    for key, fd in flat_dicts.items():
        if fd is None:
            continue
        out[key] = {"flat": fd.flat.detach().cpu(), "entries": list(fd.layout.entries)}
    torch.save(out, path)


def load_cl_state(path, flat):
    """-> dict of FlatDict on `flat`'s device; refuses a file whose tensor table differs from the model's."""
    from .cl import FlatDict
    raw = torch.load(path, map_location="cpu")
    res = {}
    for key, rec in raw.items():
        if [tuple(e) if not isinstance(e, tuple) else e for e in rec["entries"]] != list(flat.entries):
            raise ValueError(f"{path}: '{key}' was saved for a different set of trainable tensors")
        res[key] = FlatDict(flat, rec["flat"].to(flat.theta.device))
    return res
