"""Developer tool: list host<->device synchronisation points inside one bench step (torch sync debug mode)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "2", "--no-cpu-baseline"]
import torch
import bench
orig_sync = torch.cuda.synchronize
state = {"n": 0}
def patched():
    state["n"] += 1
    if state["n"] == 2:      # after warmup: from here on report implicit syncs
        torch.cuda.set_sync_debug_mode("warn")
    return orig_sync()
torch.cuda.synchronize = patched
warnings.simplefilter("always")
bench.main()
