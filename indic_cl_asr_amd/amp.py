"""`config.mixed_precision` on MI355X (R/config.yaml:10, R/cl_baseline.py:181-196).

The reference's mixed-precision branch is fp16 autocast + `torch.amp.GradScaler` (loss scaled by 2^16, gradients unscaled
before the optimizer step, the EWC penalty pre-loaded into `.grad` NOT scaled: R/cl_baseline_ewc.py:229-238, SURVEY quirk
list).  The MI355X path's reduced-precision mode is `compute_dtype="bf16"` (bf16 projections with fp32 accumulation, fp32
norms / softmax / residual stream, f16 joint lattice with its own power-of-two gradient scale kappa chosen per step):
bf16 has fp32's exponent range, so no loss scaling is needed and none is applied.  Mapping:

    mixed_precision: true   -> model_config(..., compute_dtype="bf16")   the HIP kernels (what bench.py times)
    mixed_precision: false  -> model_config(..., compute_dtype="fp32")   exact fp32 arithmetic (ATen composition + HIP losses)

`GradScaler` / `autocast` below keep the branch's CALLS working unchanged (`scaler.scale(loss).backward()`,
`scaler.step(optimizer)`, `scaler.update()`, `with autocast(device_type="cuda", enabled=...)`) with scale 1: the penalty
and the loss gradient then carry the same scale, i.e. the arithmetic of the non-AMP branch.  Do not pass a real
torch.amp.GradScaler: a 2^16 loss scale overflows the f16 lattice gradient of the fused joint (kappa is sized for the
unscaled loss); `GradScaler(init_scale=s)` here rejects s != 1 for that reason.
"""
from contextlib import nullcontext


def compute_dtype_from(config) -> str:
    """'bf16' | 'fp32' for indic_cl_asr_amd.config.model_config(compute_dtype=...) from the scripts' config.yaml."""
    mp = config.get("mixed_precision", False) if isinstance(config, dict) else getattr(config, "mixed_precision", False)
    return "bf16" if bool(mp) else "fp32"


def autocast(device_type="cuda", enabled=True, dtype=None, cache_enabled=None):
    """Stand-in for torch.amp.autocast around `training_step`: the model casts its own operands (it enters a bf16 autocast
    region around its ATen pieces itself when compute_dtype == 'bf16'), so the outer context has nothing to do."""
    return nullcontext()


class GradScaler:
    """torch.amp.GradScaler's calls as the mixed-precision branch makes them, with scale 1 (see the module docstring)."""

    def __init__(self, device="cuda", init_scale=1.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        if float(init_scale) != 1.0:
            raise ValueError("bf16 needs no loss scaling and the fused joint's f16 gradient cannot take one: init_scale must be 1")
        self._enabled = enabled

    def scale(self, outputs):
        return outputs

    def unscale_(self, optimizer):
        return None

    def step(self, optimizer, *args, **kwargs):
        return optimizer.step(*args, **kwargs)

    def update(self, new_scale=None):
        return None

    def get_scale(self):
        return 1.0

    def is_enabled(self):
        return self._enabled

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state):
        return None
