"""`import NeMo.nemo.collections.asr as nemo_asr` -> `nemo_asr.models.ASRModel.from_pretrained(...)` (R/cl_baseline.py:13,122)."""
from . import models  # noqa: F401
