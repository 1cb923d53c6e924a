"""Data-parallel step on the GPU with two ranks -- one rank per device over RCCL (backend "nccl") when the box shows two
or more GPUs, otherwise the rehearsal of the same path with both ranks on cuda:0 over gloo: the deferred optimizer update
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


def _init(rank, world, port):
    """RCCL with one rank per device when the box has enough devices (device_count() does not initialise the GPU), else
    gloo with every rank on cuda:0 (one-GPU boxes: the collectives then go through host memory, the HIP path is the same)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if torch.cuda.device_count() >= world:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        return "nccl"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return "gloo"


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
    _init(rank, world, port)
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


# ---------------------------------------------------------------------------------------------------------------------
# SyncBatchNorm (torch.nn.SyncBatchNorm.convert_sync_batchnorm, R/cl_baseline.py:133) on the HIP encoder paths: two ranks,
# each with HALF of a batch, must reproduce the single-process encoder on the UNION batch (plain BatchNorm) -- outputs of
# the frozen prefix and of the trainable blocks, the blocks' input / parameter gradients (summed over the ranks), and the
# running statistics.
def _sync_setup():
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    torch.manual_seed(0)
    cfg = model_config('tiny', d_model=128, n_layers=3, n_heads=2, pred_hidden=64, joint_hidden=64, languages=['hi', 'ta'],
                       vocab_per_lang=16, fused_batch_size=2, compute_dtype='bf16', dither=0.0)
    m = EncDecHybridRNNTCTCModel(cfg).disable_dropout().cuda().train()
    with torch.no_grad():
        for l in m.encoder.layers:
            l.conv.batch_norm.weight.uniform_(0.5, 1.5); l.conv.batch_norm.bias.normal_(0, 0.2)
    m.spec_augment_enabled = False
    freeze_layer(m, 0); m.encoder.encoder_frozen_till = 1       # layer 0 in the no-grad prefix, layers 1..2 trainable
    g = torch.Generator().manual_seed(5)
    L = 16000
    sig = torch.randn(4, L, generator=g) * 0.1
    sl = torch.tensor([L, 12000, L, 9000])
    for i in range(4):
        sig[i, sl[i]:] = 0
    R = torch.randn(4, 128, 101, generator=g)
    return m, sig, sl, R


def _encoder_pass(m, sig, sl, R):
    enc, elen = m(input_signal=sig.cuda(), input_signal_length=sl.cuda())
    T = enc.shape[2]
    valid = (torch.arange(T, device="cuda")[None, :] < elen[:, None]).unsqueeze(1)
    (enc.float() * R[:, :, :T].cuda() * valid).sum().backward()
    grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None and n.startswith("encoder.")}
    bn = m.encoder.layers[0].conv.batch_norm, m.encoder.layers[2].conv.batch_norm
    return enc.detach().float(), elen, grads, [b.running_var.clone() for b in bn], [int(b.num_batches_tracked) for b in bn]


def _sync_worker(rank, world, port, q):
    _init(rank, world, port)
    m, sig, sl, R = _sync_setup()
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
    assert isinstance(m.encoder.layers[0].conv.batch_norm, torch.nn.SyncBatchNorm)
    half = slice(2 * rank, 2 * rank + 2)
    enc, elen, grads, rvs, nbt = _encoder_pass(m, sig[half], sl[half], R[half])
    names = sorted(grads)
    flat = torch.cat([grads[n].reshape(-1) for n in names])
    dist.all_reduce(flat)                                      # parameter gradients: sum over the ranks
    # (numpy arrays are pickled by value: torch tensors travel as shared-memory handles that die with this process)
    q.put((rank, enc.cpu().numpy(), elen.cpu().numpy(), names, flat.cpu().numpy(), [v.cpu().numpy() for v in rvs], nbt))
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_single_process_union_batch():
    m, sig, sl, R = _sync_setup()
    enc1, elen1, grads1, rvs1, nbt1 = _encoder_pass(m, sig, sl, R)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    res = [(r[0], torch.from_numpy(r[1]), torch.from_numpy(r[2]), r[3], torch.from_numpy(r[4]), [torch.from_numpy(v) for v in r[5]], r[6])
           for r in res]
    enc2 = torch.cat([res[0][1], res[1][1]], 0)
    T = enc1.shape[2]
    valid = (torch.arange(T)[None, :] < elen1.cpu()[:, None]).unsqueeze(1)
    scale = (enc1.cpu() * valid).abs().max().item()
    assert torch.equal(torch.cat([res[0][2], res[1][2]]), elen1.cpu())
    assert ((enc2 - enc1.cpu()) * valid).abs().max().item() <= 2e-2 * scale      # per-rank BatchNorm would be off by O(1)
    names, flat2 = res[0][3], res[0][4]
    flat1 = torch.cat([grads1[n].reshape(-1) for n in names]).cpu()
    off = 0
    for n in names:
        k = grads1[n].numel()
        a, b = flat2[off:off + k], flat1[off:off + k]
        off += k
        if n.endswith("depthwise_conv.bias") or n.endswith("linear_k.bias"):
            continue
        rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
        assert rel < 5e-2, (n, rel)
    for r in res:                                              # running statistics: the GLOBAL batch on every rank
        for v2, v1 in zip(r[5], rvs1):
            assert torch.allclose(v2, v1.cpu(), rtol=2e-2, atol=1e-4)
        assert r[6] == nbt1


# ---------------------------------------------------------------------------------------------------------------------
# The wrapping the CL scripts apply (R/cl_baseline.py:133-134): SyncBatchNorm.convert_sync_batchnorm, then
# DistributedDataParallel(model, device_ids=[local_rank]), then `model.module.training_step` (:190) with ONE optimizer over
# model.parameters().  The model's parameters are views of cl.FlatParams' flat buffers: they must survive both wraps, the
# HIP paths must stay selected, and the ranks must end with identical weights (with the fp32 and the bf16 exchange).
def _ddp_worker(rank, world, port, q):
    try:
        backend = _init(rank, world, port)
        from indic_cl_asr_amd import cl
        from indic_cl_asr_amd.config import model_config
        from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
        out = {}
        for exch in (None, "bf16"):
            torch.manual_seed(50 + rank)            # DIFFERENT initial weights per rank: DDP's constructor broadcasts rank 0's
            cfg = model_config('tiny', d_model=128, n_layers=3, n_heads=2, pred_hidden=64, joint_hidden=64, languages=['hi', 'ta'],
                               vocab_per_lang=16, fused_batch_size=2, compute_dtype='bf16', dither=0.0)
            m = EncDecHybridRNNTCTCModel(cfg).disable_dropout().cuda()
            m.spec_augment_enabled = False
            freeze_layer(m, 0); m.encoder.encoder_frozen_till = 0
            m.ctc_wer.log_prediction = False; m.wer.log_prediction = False          # R/cl_baseline.py:127-128
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
            dev = torch.cuda.current_device()
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev], output_device=dev)
            model.train()
            opt = cl.FusedAdamW(model, lr=1e-2, grad_exchange_dtype=exch)
            assert isinstance(model.module.encoder.layers[0].conv.batch_norm, torch.nn.SyncBatchNorm)
            th0 = cl.get_params_clone(model).flat.clone()
            losses = []
            for step in range(3):
                batch = _batch(rank, step)
                opt.zero_grad()
                loss, monitor = model.module.training_step(batch, ['hi'] * len(batch[0]), compute_wer=(step == 2))
                loss.backward()
                opt.step()
                losses.append(monitor['train_loss'])
            wer = float(monitor['training_batch_wer'])
            th = cl.get_params_clone(model).flat.clone()
            both = [torch.empty_like(th) for _ in range(world)]
            dist.all_gather(both, th)
            both0 = [torch.empty_like(th0) for _ in range(world)]
            dist.all_gather(both0, th0)
            wers = [torch.zeros(1, device="cuda") for _ in range(world)]
            dist.all_gather(wers, torch.tensor([wer], device="cuda"))
            views = dict(model.module.named_parameters())
            out[str(exch)] = dict(
                same_start=bool(torch.equal(both0[0], both0[1])), same_end=bool(torch.equal(both[0], both[1])),
                moved=float((th - th0).abs().max()), finite=all(l == l for l in losses),
                views=all(views[n].data_ptr() == opt.flat.params_dict()[n].data_ptr() for n in opt.flat.names),
                wer_same=bool(torch.equal(wers[0], wers[1])), wer=wer, steps=opt.step_count, bytes=opt.exchange_bytes,
                numel=opt.flat.numel)
        q.put((rank, backend, out, None))
    except Exception:
        import traceback
        q.put((rank, "?", None, traceback.format_exc()))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_syncbn_plus_ddp_wrapped_model_trains_identically_on_two_ranks():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, backend, out, err in res:
        assert err is None, err
        for exch, r in out.items():
            assert r["same_start"], (exch, "DDP did not broadcast rank 0's weights into the flat buffer")
            assert r["same_end"], (exch, "ranks diverged")
            assert r["moved"] > 0 and r["finite"] and r["views"] and r["steps"] == 3, (exch, r)
            assert r["wer_same"] and r["wer"] == r["wer"], (exch, r)     # the step's WER is the cross-rank rate on every rank
            assert r["bytes"] == r["numel"] * (2 if exch == "bf16" else 4), (exch, r)
    for p in ps:
        assert p.exitcode == 0
