"""N>1 path on CPU: 2 ranks over gloo (127.0.0.1).  Checks the host logic of the data-parallel exchange -- one
all-reduce of the flat gradient buffer, and Fisher / omega reduced to the 1-process result over the union of shards."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    torch.manual_seed(0)
    m = EncDecHybridRNNTCTCModel(model_config('tiny'))
    freeze_layer(m, 0)
    flat = cl.FlatParams(m)
    # identical layout on every rank; views alias the flat buffers
    assert flat.params_dict()["joint.enc.weight"].data_ptr() == dict(m.named_parameters())["joint.enc.weight"].data_ptr()
    opt = cl.FusedAdamW(flat, lr=1e-3)
    flat.grad.copy_(torch.arange(flat.numel, dtype=torch.float32) * (rank + 1))
    scale = opt.allreduce_grads()
    ok1 = torch.allclose(flat.grad * scale, torch.arange(flat.numel, dtype=torch.float32) * 1.5) and scale == 0.5
    # Fisher: rank-local sums over 3 (rank 0) and 5 (rank 1) samples -> global mean over 8 samples
    g = torch.Generator().manual_seed(100 + rank)
    fish = flat.zeros()
    n_local = 3 if rank == 0 else 5
    local_sum = torch.rand(flat.numel, generator=g)
    fish.flat.copy_(local_sum)
    sums = [torch.zeros_like(local_sum) for _ in range(world)]
    dist.all_gather(sums, local_sum)
    main = cl.fisher_finish(None, fish, total_ds=n_local, e_gamma=1.0)
    ok2 = torch.allclose(main.flat, (sums[0] + sums[1]) / 8.0)
    # second task: main = gamma * main + fish
    fish2 = flat.zeros(); fish2.flat.fill_(float(rank + 1))
    main2 = cl.fisher_finish(main, fish2, total_ds=1, e_gamma=0.5)
    ok3 = torch.allclose(main2.flat, 0.5 * (sums[0] + sums[1]) / 8.0 + 1.5)
    om = flat.zeros(); om.flat.fill_(float(2 * rank + 1))
    ok4 = torch.allclose(cl.importance_finish(om, n_batches=2).flat, torch.full((flat.numel,), 1.0))
    q.put((rank, bool(ok1), bool(ok2), bool(ok3), bool(ok4)))
    dist.destroy_process_group()


def test_two_rank_gloo_exchange():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


# ---------------------------------------------------------------------------------------------------------------------
# bf16-compressed gradient exchange, the WER metric's cross-rank sum (A/metrics/wer.py: sync on compute) and a
# DistributedDataParallel-wrapped model driven the way the CL scripts drive it (R/cl_baseline.py:133-134,190: wrap, then
# `model.module.training_step`, shared optimizer over model.parameters()).
def _worker2(rank, world, port, q):
    try:
        _worker2_body(rank, world, port, q)
    except Exception:   # report instead of letting the parent wait for its queue timeout
        import traceback
        q.put((rank, False, traceback.format_exc()))
        raise


def _worker2_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from indic_cl_asr_amd import cl
    from indic_cl_asr_amd.config import model_config
    from indic_cl_asr_amd.model import EncDecHybridRNNTCTCModel, freeze_layer
    torch.manual_seed(0)
    m = EncDecHybridRNNTCTCModel(model_config('tiny', compute_dtype='fp32', dither=0.0)).disable_dropout()
    m.spec_augment_enabled = False
    freeze_layer(m, 0)
    flat = cl.FlatParams(m)
    # --- bf16 exchange: every rank ends with the SAME bf16-rounded sum, within bf16 rounding of the exact one
    g = torch.Generator().manual_seed(7 + rank)
    local = torch.randn(flat.numel, generator=g)
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    opt16 = cl.FusedAdamW(flat, lr=1e-3, grad_exchange_dtype="bf16", defer_update=False)
    flat.grad.copy_(local)
    scale = opt16.allreduce_grads()
    exact = both[0] + both[1]
    got = flat.grad.clone()
    mine = [torch.zeros_like(got) for _ in range(world)]
    dist.all_gather(mine, got)
    ok_same = bool(torch.equal(mine[0], mine[1])) and scale == 0.5 and opt16.exchange_bytes == flat.numel * 2
    ref16 = both[0].bfloat16().float() + both[1].bfloat16().float()
    ok_close = bool(((got - exact).abs() <= 2.0 ** -7 * (both[0].abs() + both[1].abs()) + 1e-6).all()) and \
        bool(((got - ref16).abs() <= 2.0 ** -8 * ref16.abs() + 1e-6).all())
    # --- WER.compute(): (scores, words) summed over the ranks in one exchange
    m.wer.scores.fill_(3 + rank); m.wer.words.fill_(10 * (rank + 1))
    wer, s, w = m.wer.compute()
    ok_wer = (float(s), float(w)) == (7.0, 30.0) and abs(float(wer) - 7.0 / 30.0) < 1e-7
    gw, gs_, gw_ = m.wer.grouped([[1, 2], [3]], [[1, 2 + rank], [3]], None, 1)      # two groups of one utterance each
    ok_grp = (float(gs_), float(gw_)) == (1.0, 6.0) and abs(float(gw) - 0.5 * (1 / 4 + 0.0)) < 1e-7
    # --- DDP wrap on CPU (the product has no CPU forward: the wrapped training_step itself runs in tests/test_dist_gpu.py).
    # What is checked here: parameters that are views of the flat buffers survive the wrap (DDP broadcasts rank 0's values
    # INTO them), names are unchanged under `.module`, and the shared optimizer over the wrapped model exchanges and updates.
    torch.manual_seed(123 + rank)                # different weights per rank on purpose
    m2 = EncDecHybridRNNTCTCModel(model_config('tiny', compute_dtype='fp32', dither=0.0)).disable_dropout()
    freeze_layer(m2, 0); m2.encoder.encoder_frozen_till = 0
    flat2 = cl.FlatParams(m2)                    # (either order works: before the wrap here, after it in the GPU test)
    ddp = torch.nn.parallel.DistributedDataParallel(m2)
    opt = cl.FusedAdamW(ddp, lr=1e-2)           # flat_of(ddp) == flat2; deferred update on its own communicator
    names = [n for n, _ in ddp.module.named_parameters()]
    ok_names = names == [n for n, _ in EncDecHybridRNNTCTCModel(model_config('tiny')).named_parameters()] and opt.flat is flat2
    th0 = cl.get_params_clone(ddp).flat.clone()
    t0s = [torch.zeros_like(th0) for _ in range(world)]
    dist.all_gather(t0s, th0)
    ok_bcast = bool(torch.equal(t0s[0], t0s[1]))                         # rank 0's weights everywhere, inside theta
    gg = torch.Generator().manual_seed(10 * rank)
    local2 = torch.randn(flat2.numel, generator=gg)
    flat2.grad.copy_(local2)                                             # stands for loss.backward()
    sc = opt.allreduce_grads()                                           # (the AdamW launch itself is HIP-only: GPU test)
    l2 = [torch.zeros_like(local2) for _ in range(world)]
    dist.all_gather(l2, local2)
    views = dict(ddp.module.named_parameters())
    ok_ddp = ok_bcast and sc == 0.5 and bool(torch.allclose(flat2.grad, l2[0] + l2[1])) and \
        all(views[n].data_ptr() == flat2.params_dict()[n].data_ptr() for n in flat2.names) and \
        all(views[n].grad.data_ptr() == flat2.grads_dict()[n].data_ptr() for n in flat2.names)
    q.put((rank, ok_same, ok_close, ok_wer, ok_grp, ok_names, ok_ddp))
    dist.destroy_process_group()


def test_two_rank_bf16_exchange_wer_sync_and_ddp_wrap():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker2, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r
