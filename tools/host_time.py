"""Developer tool: host-side (enqueue) time of the pieces of one bench step, GPU kept saturated (no syncs)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from indic_cl_asr_amd import cl, model as M, encoder as E
from indic_cl_asr_amd.ops import block as BK

acc = collections.defaultdict(float)
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] += time.perf_counter() - t; return r
    setattr(obj, name, g)

wrap(E.ConformerEncoder, "_fast_prefix", "enc.frozen_prefix")
wrap(E.ConformerEncoder, "forward", "enc.total")
wrap(BK._ConformerBlockFn, "forward", "blk.fwd")
wrap(BK._ConformerBlockFn, "backward", "blk.bwd")
wrap(M.EncDecHybridRNNTCTCModel, "training_step", "training_step")
wrap(M.EncDecHybridRNNTCTCModel, "forward", "model.forward(pre+enc)")
wrap(cl.FusedAdamW, "step", "opt.step")
wrap(cl, "ewc_penalty_into_grads", "ewc_penalty")
wrap(torch.Tensor, "backward", "loss.backward")
sys.argv = [sys.argv[0], "--steps", "20", "--warmup", "5", "--no-cpu-baseline"] + sys.argv[1:]
orig = torch.cuda.synchronize
n = {"c": 0}
def sync():
    n["c"] += 1
    if n["c"] == 1: acc.clear()
    return orig()
torch.cuda.synchronize = sync
bench.main()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {v / 20 * 1e3:7.3f} ms/step", file=sys.stderr)
