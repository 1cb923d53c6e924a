"""ctypes binding of libindicasr_hip.so (the C ABI declared in include/indicasr.h).

The product path has NO fallback: if the shared object is missing or a symbol is absent,
importing/using an op raises.  Status codes are turned into exceptions the way the reference turns
RNNTStatus into RuntimeError (NeMo/nemo/collections/asr/parts/numba/rnnt_loss/rnnt.py:84-85,115-116).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IA_LIB_PATH") or os.path.join(_HERE, "libindicasr_hip.so")  # IA_LIB_PATH: developer override

IA_OK = 0
_STATUS = {-1: "IA_INVALID_VALUE", -2: "IA_WORKSPACE_TOO_SMALL", -3: "IA_LAUNCH_FAILED", -4: "IA_UNSUPPORTED"}

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t
_i64 = _c.c_int64

# name -> (restype, argtypes); must list every symbol include/indicasr.h declares (tests/test_abi.py checks)
class BlockParams(_c.Structure):
    """ia_block_params of include/indicasr.h (device pointers of one Conformer block for the native prefix executor)."""
    _fields_ = ([(n, _c.c_void_p) for n in (
        "w_ff1a", "w_ff1b", "w_qkv", "w_pos", "w_out", "w_pw1", "w_pw2", "w_ff2a", "w_ff2b",
        "b_ff1a", "b_ff1b", "b_qkv", "b_out", "b_pw1", "b_pw2", "b_ff2a", "b_ff2b",
        "ln_ff1_g", "ln_ff1_b", "ln_att_g", "ln_att_b", "ln_conv_g", "ln_conv_b", "ln_ff2_g", "ln_ff2_b", "ln_out_g", "ln_out_b",
        "pos_u", "pos_v", "dw_w", "dw_b", "bn_g", "bn_b", "bn_rm", "bn_rv", "bn_nbt")]
        + [(n, _c.c_float) for n in ("ln_eps", "bn_eps", "bn_momentum", "p_drop", "p_ff", "p_att", "fc_factor")]
        + [(n, _c.c_int) for n in ("d", "d_ff", "n_heads", "ksz")]
        + [("pl_cached", _c.c_void_p), ("w_pw1_glu", _c.c_void_p), ("b_pw1_glu", _c.c_void_p)])


class BlockSaved(_c.Structure):
    """ia_block_saved of include/indicasr.h (activations a trainable block keeps from forward to backward)."""
    _fields_ = [(n, _c.c_void_p) for n in ("y1", "h1p", "h1", "x1", "y2", "qkv", "pl", "ctxv", "x2", "y3", "c2", "z", "sums",
                                           "c3", "x3", "y4", "h4p", "h4", "x4", "lse")]


class TnProblem(_c.Structure):
    """ia_tn_problem of include/indicasr.h (one weight gradient of a grouped TN GEMM launch)."""
    _fields_ = [("dY", _c.c_void_p), ("X", _c.c_void_p), ("dW", _c.c_void_p), ("db", _c.c_void_p),
                ("ldy", _c.c_int), ("ldx", _c.c_int), ("M", _c.c_int), ("n", _c.c_int), ("k", _c.c_int)]


class BlockGrads(_c.Structure):
    """ia_block_grads of include/indicasr.h (where a trainable block's parameter gradients are written)."""
    _fields_ = [(n, _c.c_void_p) for n in (
        "w_ff1a", "b_ff1a", "w_ff1b", "b_ff1b", "w_qkv", "b_qkv", "w_pos", "w_out", "b_out", "w_pw1", "b_pw1", "w_pw2", "b_pw2",
        "w_ff2a", "b_ff2a", "w_ff2b", "b_ff2b",
        "ln_ff1_g", "ln_ff1_b", "ln_att_g", "ln_att_b", "ln_conv_g", "ln_conv_b", "ln_ff2_g", "ln_ff2_b", "ln_out_g", "ln_out_b",
        "dw_w", "dw_b", "bn_g", "bn_b")]


SIGNATURES = {
    "ia_version": (_c.c_char_p, []),
    "ia_gemm_bf16_ex2": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _f, _c.c_uint, _f, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "ia_transpose16_multi": (_i, [_vp, _i, _vp]),
    "ia_swap01_cast": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "ia_ctc_row_lse": (_i, [_vp, _i, _i64, _i, _vp, _vp]),
    "ia_ctc_forward_logits": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ia_ctc_backward_logits": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _f, _vp, _i, _vp, _sz, _vp]),
    "ia_select_rows_cast": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp]),
    "ia_rows_scatter_add": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "ia_loss_combine": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ia_loss_combine_bwd": (_i, [_vp, _i, _f, _vp, _vp, _vp]),
    "ia_embed_sos": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ia_embed_sos_bwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "ia_multi_axpy": (_i, [_vp, _i, _i, _vp]),
    "ia_peak_stream_copy": (_i, [_vp, _vp, _sz, _vp]),
    "ia_peak_mfma_bf16": (_i, [_vp, _i, _i, _vp]),
    "ia_peak_mfma_bf16_flops": (_c.c_double, [_i, _i]),
    "ia_rnnt_workspace_bytes": (_sz, [_i, _i, _i]),
    "ia_rnnt_loss": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "ia_rnnt_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "ia_rnnt_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "ia_rnnt_export_alphas_betas": (_i, [_vp, _sz, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ia_joint_ld": (_i, [_i]),
    "ia_joint_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _c.c_uint, _vp, _i, _vp, _sz, _vp]),
    "ia_joint_fwd_box": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _c.c_uint, _vp, _i, _vp, _sz, _vp]),
    "ia_joint_extra_scratch_elems": (_i64, []),
    "ia_joint_extra_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ia_joint_extra_grad": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ia_lattice_add_f16": (_i, [_vp, _vp, _i64, _vp]),
    "ia_relpos_attention_flash_lse": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp, _vp]),
    "ia_relpos_attention_flash_bwd_dims": (_i, [_i, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "ia_relpos_attention_flash_bwd_ws_elems": (_i64, [_i, _i, _i, _i]),
    "ia_relpos_attention_flash_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp, _i, _vp,
                                           _vp, _vp, _vp, _vp, _vp]),
    "ia_quantize_fp8_rows": (_i, [_vp, _i, _i, _i64, _i, _vp, _i, _vp, _vp]),
    "ia_gemm_fp8": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _f, _c.c_uint, _f, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "ia_gemm_tn_grouped_scratch_elems": (_i64, [_vp, _i]),
    "ia_gemm_tn_bf16_grouped": (_i, [_vp, _i, _vp, _vp]),
    "ia_greedy_decode_lds_bytes": (_i, [_i, _i, _i]),
    "ia_greedy_rnnt_decode": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "ia_greedy_rnnt_decode_bf16w": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "ia_edit_distance_batch": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "ia_greedy_decode_scratch_bytes": (_sz, [_i, _i, _i]),
    "ia_greedy_decode_cluster": (_i, [_i, _i, _i]),
    "ia_greedy_rnnt_decode_bf16w_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "ia_quantize_mxfp8": (_i, [_vp, _i, _i, _i64, _i, _vp, _i, _vp, _i, _vp]),
    "ia_gemm_mxfp8": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _f, _c.c_uint, _f, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "ia_cast_pad_bf16": (_i, [_vp, _i, _i64, _i, _vp, _i, _vp]),
    "ia_rnnt_lattice": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "ia_joint_backward_g": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _f, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp,
                                 _vp]),
    "ia_joint_backward_g_skip": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _f, _vp, _i, _i, _vp, _vp, _vp, _sz, _i, _vp,
                                      _vp, _vp]),
    "ia_joint_backward_g_dbias_scratch_elems": (_i64, [_i]),
    "ia_joint_hidden_t": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _c.c_uint, _vp]),
    "ia_joint_hidden": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _c.c_uint, _vp]),
    "ia_joint_dh_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _c.c_uint, _vp, _vp]),
    "ia_joint_dh_reduce_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "ia_joint_dh_fused_supported": (_i, [_i, _i, _i]),
    "ia_joint_dh_k": (_i, []),
    "ia_joint_dh_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _c.c_uint, _vp, _vp]),
    "ia_joint_dh_fused_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _c.c_uint, _vp, _vp]),
    "ia_joint_dh_fused_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "ia_joint_dw_fused_supported": (_i, [_i, _i, _i]),
    "ia_joint_dw_fused_scratch_elems": (_i64, [_i, _i, _i, _i, _i]),
    "ia_joint_dw_fused": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp, _vp]),
    "ia_gemm_bf16_ln_supported": (_i, [_i, _i]),
    "ia_gemm_bf16_ln": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _f, _c.c_uint, _f, _vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp]),
    "ia_joint_dw_fused_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp, _vp]),
    "ia_gemm_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _f, _c.c_uint, _f, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "ia_gemm_bf16_ex": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _f, _c.c_uint, _f, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "ia_subsample_conv1": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ia_subsample_conv2": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "ia_layernorm": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "ia_glu_dwconv": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ia_dwconv_scratch_elems": (_i64, [_i, _i, _i, _i]),
    "ia_colsum_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ia_dwconv_time": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "ia_dwconv_time_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ia_dwconv_glu_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ia_dwconv_glu_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ia_bn_silu": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp]),
    "ia_gemm_bnsilu_supported": (_i, [_i]),
    "ia_gemm_bnsilu_bf16": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _i, _i, _i, _i, _vp, _f, _c.c_uint, _f,
                                 _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "ia_gemm_bnsilu_bf16_keep": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _i, _i, _i, _i, _vp, _f, _c.c_uint,
                                      _f, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "ia_glu_dwconv_fixed": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ia_dwconv_gated_fixed": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ia_attn_vt_elems": (_sz, [_i, _i, _i]),
    "ia_relpos_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp, _vp]),
    "ia_relpos_attention_flash_supported": (_i, [_i, _i]),
    "ia_relpos_attention_flash": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp]),
    "ia_relpos_attention_bwd_dims": (_i, [_i, _vp, _vp, _vp]),
    "ia_relpos_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _c.c_uint, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp]),
    "ia_attn_bwd_unpack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ia_attn_bwd_unpack_scratch_elems": (_i64, [_i, _i, _i]),
    "ia_lstm_scratch_bytes": (_sz, [_i, _i]),
    "ia_lstm_lds_bytes": (_i, [_i, _i]),
    "ia_lstm_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "ia_lstm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "ia_feat_frames": (_i, [_vp, _i, _i, _i, _i, _i, _f, _f, _c.c_uint, _vp, _i, _vp]),
    "ia_gemm_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "ia_feat_power": (_i, [_vp, _i64, _i, _i, _i, _vp, _i, _vp]),
    "ia_feat_logmel_t": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "ia_feat_preemph": (_i, [_vp, _i, _i, _f, _f, _c.c_uint, _vp, _vp]),
    "ia_feat_logmel_fft_supported": (_i, [_i, _i, _i, _i]),
    "ia_feat_logmel_fft": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "ia_feat_normalize": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp]),
    "ia_ctc_workspace_bytes": (_sz, [_i, _i, _i]),
    "ia_ctc_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ia_ctc_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ia_gemm_tn_scratch_elems": (_i64, [_i, _i, _i]),
    "ia_gemm_tn_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ia_ffn_fused_supported": (_i, [_i, _i]),
    "ia_ffn_fused": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _c.c_uint, _f, _c.c_uint, _vp, _vp, _vp, _i, _vp]),
    "ia_ffn_fused_tail_supported": (_i, [_i, _i, _i]),
    "ia_ffn_fused_tail": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _c.c_uint, _f, _c.c_uint, _vp, _vp, _vp, _i,
                               _vp, _vp, _vp, _i, _vp]),
    "ia_conformer_prefix_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "ia_conformer_prefix_ws_sums_offset": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "ia_conformer_prefix_fwd_seg": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _c.c_uint, _c.c_uint, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ia_conformer_prefix_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _c.c_uint, _c.c_uint, _i, _vp, _sz, _vp]),
    "ia_conformer_block_supported": (_i, [_i, _i, _i, _i, _i]),
    "ia_conformer_block_bwd_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ia_conformer_block_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _c.c_uint, _vp, _vp, _vp, _vp, _vp]),
    "ia_conformer_block_fwd_phase": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _c.c_uint, _vp, _vp, _vp, _vp, _i, _vp]),
    "ia_conformer_block_bwd_a": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _c.c_uint, _vp, _sz, _vp, _vp, _vp]),
    "ia_conformer_block_bwd_a_phase": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _c.c_uint, _vp, _sz, _vp, _vp, _i, _vp, _vp]),
    "ia_conformer_block_bwd_b": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _c.c_uint, _vp, _sz, _vp, _vp, _i, _vp]),
    "ia_layernorm_bwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "ia_layernorm_bwd_drop": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _f, _vp, _vp, _i, _vp, _vp, _f, _f, _c.c_uint, _vp, _i, _vp, _vp]),
    "ia_layernorm_bwd_partial_rows": (_i, [_i]),
    "ia_partials_finish_multi": (_i, [_vp, _i, _vp]),
    "ia_layernorm_bwd_scratch_elems": (_i64, [_i, _i]),
    "ia_silu_dropout": (_i, [_vp, _i64, _i, _f, _c.c_uint, _vp, _vp]),
    "ia_silu_dropout_bwd": (_i, [_vp, _vp, _i64, _i, _f, _c.c_uint, _vp, _vp]),
    "ia_scale_dropout_bf16": (_i, [_vp, _i64, _i, _f, _f, _c.c_uint, _vp, _vp]),
    "ia_bn_silu_bwd": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "ia_bn_silu_bwd_scratch_elems": (_i64, [_i64, _i]),
    "ia_bn_sync_finish": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _f, _vp]),
    "ia_bn_silu_bwd_reduce": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "ia_bn_silu_bwd_apply": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "ia_glu_mask": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ia_glu_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ia_attn_keepmask": (_i, [_i, _i, _i, _f, _c.c_uint, _vp, _vp]),
    "ia_cl_chunk_elems": (_i, []),
    "ia_cl_penalty": (_i, [_vp, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "ia_cl_fisher_accumulate": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "ia_cl_abs_accumulate": (_i, [_vp, _vp, _i64, _vp]),
    "ia_adamw_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    "ia_adamw_step_segmented": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _f, _f, _f, _f, _f, _vp, _vp]),
}

_lib = None


def build(verbose=False):
    """Compile every .hip under csrc/ for gfx950 into libindicasr_hip.so (in-tree)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc")], stdout=out)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the product path and has no fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C indic_cl_asr_amd/csrc`.")
        # torch first: its wheel bundles a libamdhip64; if this library were loaded before it, its kernels would register
        # with the system HIP runtime while torch's streams and allocations live in the bundled one (every launch on a
        # torch stream then fails: seen as IA_LAUNCH_FAILED when build() and smoke() ran in one process)
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is absent
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(status, what):
    if status != IA_OK:
        raise RuntimeError(f"{what} failed with status {status} ({_STATUS.get(status, 'unknown')})")


def version():
    return lib().ia_version().decode()


def stream_ptr():
    """Raw hipStream_t of torch's current stream (kernels are enqueued there, never synchronised)."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None
