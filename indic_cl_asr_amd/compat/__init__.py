"""Import-path aliases for the reference's driver scripts.

R/cl_baseline{,_ewc,_mas,_lwf}.py begin with

    import NeMo.nemo.collections.asr as nemo_asr
    from NeMo.nemo.collections.asr.models.hybrid_rnnt_ctc_models import TranscribeConfig, InternalTranscribeConfig

(R/cl_baseline.py:13-14) and build the model with `nemo_asr.models.ASRModel.from_pretrained(name)` (:122).  The directory
next to this file holds a package tree of those NAMES whose modules re-export this package's classes -- no reference text,
nothing of NeMo's.  `install()` puts the tree on `sys.path`, after which the scripts' import lines resolve unchanged:

    import indic_cl_asr_amd.compat as compat; compat.install()       # the one line a maintainer adds at the top

Pretrained weights cannot be fetched here (the reference pulls them from the HF hub by model NAME): `ASRModel.from_pretrained`
looks for `<name with '/' replaced by '__'>.nemo` (or `.pth` + a preset) under `$IA_PRETRAINED_DIR` and fails loudly
otherwise; `ASRModel.restore_from(path)` takes a local `.nemo` archive.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def install():
    """Make `NeMo.nemo.collections.asr...` importable (idempotent).  A real NeMo checkout that is already importable under
    the same top-level name wins -- this never shadows it."""
    if "NeMo" in sys.modules and not getattr(sys.modules["NeMo"], "__ia_alias__", False):
        return False
    if _HERE not in sys.path:
        sys.path.append(_HERE)
    return True
