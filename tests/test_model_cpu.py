"""CPU checks of the product's host logic: module tree / parameter names / counts, config parsing, frame counts.
(No HIP compute here; numerics parity with the oracle runs on the GPU in tests/test_step_gpu.py.)"""
import os

import pytest
import torch

from indic_cl_asr_amd.config import load_config, model_config, override_config_with_args
from indic_cl_asr_amd.encoder import subsampled_length
from indic_cl_asr_amd.features import mel_filterbank_slaney, mel_frame_count
from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
from oracle import step_ref as S


def test_parameter_tree_matches_oracle_and_reference_counts():
    m = EncDecHybridRNNTCTCModel(model_config('tiny', compute_dtype='fp32'))
    o = S.OracleHybridModel(d_model=32, n_layers=2, n_heads=4, pred_hidden=24, joint_hidden=24, languages=['hi', 'ta'],
                            vocab_per_lang=16, fused_batch_size=2)
    a = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert a == b
    # the checkpoint the reference fine-tunes: 129.250967 M total / 115.111424 M encoder / 4.362774 M joint
    # (R/dataset_gen.ipynb cells 16, 22)
    big = EncDecHybridRNNTCTCModel(model_config('ai4b_large'))
    n = lambda mod: sum(p.numel() for p in mod.parameters())
    assert n(big) == 129250967 and n(big.encoder) == 115111424 and n(big.joint) == 4362774
    names = dict(big.named_parameters())
    for k in ("encoder.pre_encode.conv.0.weight", "encoder.layers.16.self_attn.pos_bias_u",
              "encoder.layers.0.conv.depthwise_conv.weight", "decoder.prediction.dec_rnn.lstm.weight_ih_l0",
              "joint.joint_net.2.hi.weight", "ctc_decoder.decoder_layers.0.weight"):
        assert k in names, k


def test_freeze_layer_semantics():
    m = EncDecHybridRNNTCTCModel(model_config('tiny', n_layers=4))
    freeze_layer(m, 1)
    m.encoder.encoder_frozen_till = 1
    req = {n: p.requires_grad for n, p in m.named_parameters()}
    assert not req["encoder.pre_encode.out.weight"] and not req["encoder.layers.1.norm_out.weight"]
    assert req["encoder.layers.2.norm_out.weight"] and req["joint.enc.weight"] and req["ctc_decoder.decoder_layers.0.bias"]


def test_frame_counts_are_bit_exact_with_oracle_rule():
    fb = S.FilterbankFeatures()
    for L in list(range(1, 3000, 7)) + [80000, 240000, 480000, 135977]:
        tm = int(fb.get_seq_len(torch.tensor(L)))
        assert mel_frame_count(L) == tm
        assert subsampled_length(tm) == int(S.calc_length(torch.tensor(tm)))


def test_mel_filterbank_equals_oracle_restatement():
    import numpy as np
    assert np.allclose(mel_filterbank_slaney(), S.slaney_mel_filterbank(), atol=1e-7)


def test_reference_config_yaml_parses_with_overrides(tmp_path):
    ref = "/root/reference/config.yaml"
    text = open(ref).read() if os.path.exists(ref) else (
        "batch_size: 16\nlearning_rate: 0.0001\nepochs: 1\ndistributed: true\nmixed_precision: false\n"
        "model:\n  freeze_encoder_till: 12\ncl_config:\n  e_lambda: 10\n  e_gamma: 1\n  mas_ctx: 0.3\n")
    p = tmp_path / "config.yaml"
    p.write_text(text)
    cfg = load_config(str(p))
    cfg = override_config_with_args(cfg, ["--cl_config.e_lambda", "5", "--mixed_precision", "true", "--batch_size", "32"])
    assert cfg.cl_config.e_lambda == 5 and cfg.mixed_precision is True and cfg.batch_size == 32
    assert cfg.model.freeze_encoder_till == 12


def test_mixed_precision_key_selects_the_compute_dtype_and_the_scaler_is_scale_one():
    """R/config.yaml `mixed_precision` -> compute dtype (amp.py), and the GradScaler calls of R/cl_baseline.py:181-196 with
    scale 1 (identity on the loss, step == optimizer.step)."""
    import pytest
    from indic_cl_asr_amd import amp
    from indic_cl_asr_amd.config import AttrDict, model_config
    assert amp.compute_dtype_from(AttrDict(mixed_precision=True)) == "bf16"
    assert amp.compute_dtype_from(AttrDict(mixed_precision=False)) == "fp32"
    assert amp.compute_dtype_from({}) == "fp32"
    assert model_config("tiny", compute_dtype=amp.compute_dtype_from({"mixed_precision": True})).compute_dtype == "bf16"
    sc = amp.GradScaler()
    x = torch.tensor(3.0, requires_grad=True)
    assert sc.scale(x) is x and sc.get_scale() == 1.0

    class Opt:
        n = 0

        def step(self):
            self.n += 1
            return "stepped"
    o = Opt()
    assert sc.step(o) == "stepped" and o.n == 1
    sc.update(); sc.unscale_(o)
    with amp.autocast(device_type="cuda", enabled=True):
        pass
    with pytest.raises(ValueError):
        amp.GradScaler(init_scale=65536.0)
