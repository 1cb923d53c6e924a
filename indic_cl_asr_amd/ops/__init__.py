"""Device ops of the hot path.

Every op here runs on the MI355X.  Ops backed by a hand-written HIP kernel go through the C ABI
(include/indicasr.h -> libindicasr_hip.so) and raise if the library is missing; the remaining ones are
compositions of ROCm ATen device kernels (hipBLASLt GEMMs, MIOpen conv/LSTM) kept as plumbing until their
kernel lands (DESIGN.md lists which is which).  Nothing in this package touches oracle/ or the CPU.
"""
from .frontend import log_mel, normalize_mask, spec_augment_  # noqa: F401
from .attention import rel_pos_attention  # noqa: F401
from .convmod import glu_dwconv_bn_silu  # noqa: F401
