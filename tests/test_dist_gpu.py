"""Data-parallel step on the GPU with two ranks (gloo over 127.0.0.1, both ranks on cuda:0): the deferred optimizer update
(gradient all-reduce launched asynchronously in FusedAdamW.step, AdamW applied right before the first trainable module of
the next forward) must leave exactly the weights of the immediate update, identical on both ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch(rank, step, B=3, L=12000, U=8):
    g = torch.Generator().manual_seed(1000 * rank + step)
    sl = torch.tensor([L] + [int(L * (0.6 + 0.4 * torch.rand(1, generator=g))) for _ in range(B - 1)])
    sig = torch.randn(B, L, generator=g) * 0.1
    tl = torch.tensor([U] + [int(torch.randint(1, U + 1, (1,), generator=g)) for _ in range(B - 1)])
    tr = torch.randint(0, 16, (B, U), generator=g)
    return tuple(t.cuda() for t in (sig, sl, tr, tl))


def _run(rank, defer):
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    torch.manual_seed(0)
    cfg = model_config('tiny', d_model=128, n_layers=3, n_heads=2, pred_hidden=64, joint_hidden=64, languages=['hi', 'ta'],
                       vocab_per_lang=16, fused_batch_size=2, compute_dtype='bf16', dither=0.0)
    m = EncDecHybridRNNTCTCModel(cfg).disable_dropout().cuda().train()
    m.spec_augment_enabled = False
    freeze_layer(m, 0); m.encoder.encoder_frozen_till = 0
    flat = cl.FlatParams(m)
    opt = cl.FusedAdamW(flat, lr=1e-2, defer_update=defer)
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss, mon = m.training_step(_batch(rank, step), ['hi'] * 3)
        loss.backward()
        opt.step()
        losses.append(mon['train_loss'])
    theta = cl.get_params_clone(m).flat.clone()      # flushes a pending update
    return theta, losses, opt.step_count


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th_now, l_now, n_now = _run(rank, defer=False)
    th_def, l_def, n_def = _run(rank, defer=True)
    same_modes = bool(torch.equal(th_now, th_def)) and l_now == l_def and n_now == n_def == 3
    other = [torch.empty_like(th_def) for _ in range(world)]
    dist.all_gather(other, th_def)
    same_ranks = bool(torch.equal(other[0], other[1]))
    moved = bool((th_def - _initial_theta()).abs().max().item() > 0)
    q.put((rank, same_modes, same_ranks, moved, l_now, l_def))
    dist.destroy_process_group()


def _initial_theta():
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    torch.manual_seed(0)
    cfg = model_config('tiny', d_model=128, n_layers=3, n_heads=2, pred_hidden=64, joint_hidden=64, languages=['hi', 'ta'],
                       vocab_per_lang=16, fused_batch_size=2, compute_dtype='bf16', dither=0.0)
    m = EncDecHybridRNNTCTCModel(cfg).cuda()
    freeze_layer(m, 0)
    return cl.FlatParams(m).theta.clone()


def test_deferred_update_matches_immediate_update_two_ranks():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1], ("deferred != immediate", r)
        assert r[2], ("ranks diverged", r)
        assert r[3], ("weights did not move", r)
