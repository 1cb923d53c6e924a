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
