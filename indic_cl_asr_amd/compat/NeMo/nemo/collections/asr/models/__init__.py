"""Model names the CL scripts reach through `nemo_asr.models` (R/cl_baseline.py:122, R/utils.py:500)."""
import os

from indic_cl_asr_amd import checkpoint as _ck
from indic_cl_asr_amd.config import PRESETS as _PRESETS
from indic_cl_asr_amd.config import model_config as _model_config
from indic_cl_asr_amd.model import EncDecHybridRNNTCTCBPEModel, EncDecHybridRNNTCTCModel  # noqa: F401


class ASRModel:
    """The two constructors the scripts use.  Both return the MI355X `EncDecHybridRNNTCTCModel`."""

    @classmethod
    def restore_from(cls, restore_path, strict=False, map_location=None, **overrides):
        model, report = _ck.model_from_nemo(restore_path, strict=strict, **overrides)
        model._ia_load_report = report
        return model.to(map_location) if map_location is not None else model

    @classmethod
    def from_pretrained(cls, model_name, strict=False, map_location=None, preset="ai4b_large", **overrides):
        """`ai4bharat/indicconformer_stt_hi_hybrid_rnnt_large` is a hub NAME in the reference; there is no network here, so the
        archive must already be on disk: $IA_PRETRAINED_DIR/<name with '/' -> '__'>.nemo, or a state dict `.pth` of the same
        stem (loaded into `preset`, default the checkpoint family the scripts fine-tune)."""
        root = os.environ.get("IA_PRETRAINED_DIR", "")
        stem = os.path.join(root, str(model_name).replace("/", "__"))
        if root and os.path.isfile(stem + ".nemo"):
            return cls.restore_from(stem + ".nemo", strict=strict, map_location=map_location, **overrides)
        if root and os.path.isfile(stem + ".pth"):
            if preset not in _PRESETS:
                raise ValueError(f"unknown preset {preset!r}")
            model = EncDecHybridRNNTCTCModel(_model_config(preset, **overrides))
            model._ia_load_report = _ck.load_weights(model, stem + ".pth", strict=strict)
            return model.to(map_location) if map_location is not None else model
        raise FileNotFoundError(
            f"from_pretrained({model_name!r}): pretrained weights are fetched from the HF hub by the reference; this build has "
            f"no network.  Put {os.path.basename(stem)}.nemo (or .pth) under $IA_PRETRAINED_DIR"
            + (f" (= {root})" if root else " (unset)") + ", or call ASRModel.restore_from(<local .nemo>).")
