"""Relative-position multi-head self-attention core (multi_head_attention.py:197-250 semantics).

scores[i,j] = ((q_i+u)·k_j + (q_i+v)·p[T-1-i+j]) / sqrt(dk): the reference's rel_shift pad/view trick
(:184-195) restated as index arithmetic on the [2T-1] relative-position axis; keys j >= len_b are excluded
(fill -10000 then zero after softmax, :108-111), and so are all keys for padded queries (their row is
uniform over the masked keys in the reference; it never reaches a loss because every consumer masks by length).
"""
import math

import torch


def _rel_shift_index(T, device):
    i = torch.arange(T, device=device).view(T, 1)
    j = torch.arange(T, device=device).view(1, T)
    return (T - 1 - i + j)  # [T,T] indices into the 2T-1 axis


def rel_pos_attention(q, k, v, p, bias_u, bias_v, lens, dropout_p=0.0, training=False, keep_mask=None):
    """q,k,v: [B,h,T,dk]; p: [h,2T-1,dk]; bias_u/bias_v: [h,dk]; lens: [B] i64 -> context [B,h,T,dk]."""
    B, h, T, dk = q.shape
    scale = 1.0 / math.sqrt(dk)
    qu = q + bias_u.view(1, h, 1, dk).to(q.dtype)
    qv = q + bias_v.view(1, h, 1, dk).to(q.dtype)
    ac = torch.matmul(qu, k.transpose(-2, -1))                       # [B,h,T,T]
    bd_full = torch.matmul(qv, p.unsqueeze(0).transpose(-2, -1))     # [B,h,T,2T-1]
    idx = _rel_shift_index(T, q.device)
    bd = torch.gather(bd_full, 3, idx.view(1, 1, T, T).expand(B, h, T, T))
    scores = (ac + bd).float() * scale
    valid = torch.arange(T, device=q.device)[None, :] < lens[:, None]            # [B,T]
    mask = ~(valid[:, :, None] & valid[:, None, :])                                # [B,T,T]
    scores = scores.masked_fill(mask.unsqueeze(1), -10000.0)
    attn = torch.softmax(scores, dim=-1).masked_fill(mask.unsqueeze(1), 0.0)
    if keep_mask is not None:      # explicit dropout mask (0 or 1/(1-p)) [B,h,T,T]: the HIP kernel's, for its backward
        attn = attn * keep_mask.to(attn.dtype)
    elif training and dropout_p > 0.0:
        attn = torch.nn.functional.dropout(attn, dropout_p, True)
    return torch.matmul(attn.to(v.dtype), v)
