"""`from NeMo.nemo.collections.asr.models.hybrid_rnnt_ctc_models import TranscribeConfig, InternalTranscribeConfig`
(R/cl_baseline.py:14, R/cl_baseline_ewc.py, _mas.py, _lwf.py, finetune.py): the dataclasses and the model class of
indic_cl_asr_amd.model under the reference's module path."""
from indic_cl_asr_amd.model import (EncDecHybridRNNTCTCModel, InternalTranscribeConfig,  # noqa: F401
                                    TranscribeConfig)
