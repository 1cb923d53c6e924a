"""The import lines of the reference's driver scripts resolve against the alias tree (SURVEY.md 8(b);
R/cl_baseline.py:13-14,122): names only -- every object is this package's."""
import dataclasses
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_lines_resolve_to_this_package(tmp_path):
    code = r'''
import sys
sys.path.insert(0, %r)
import indic_cl_asr_amd.compat as compat
assert compat.install() is True and compat.install() is True          # idempotent
import NeMo.nemo.collections.asr as nemo_asr                                                     # R/cl_baseline.py:13
from NeMo.nemo.collections.asr.models.hybrid_rnnt_ctc_models import TranscribeConfig, InternalTranscribeConfig   # :14
import indic_cl_asr_amd.model as M
assert TranscribeConfig is M.TranscribeConfig and InternalTranscribeConfig is M.InternalTranscribeConfig
assert nemo_asr.models.EncDecHybridRNNTCTCBPEModel is M.EncDecHybridRNNTCTCModel
cfg = TranscribeConfig(batch_size=16, return_hypotheses=False, num_workers=0, verbose=False, logprobs=True, language_id="hi")
cfg._internal = InternalTranscribeConfig()                                                       # R/cl_baseline.py:162-172
cfg._internal.temp_dir = "/tmp/x"
try:
    nemo_asr.models.ASRModel.from_pretrained("ai4bharat/indicconformer_stt_hi_hybrid_rnnt_large")
    raise SystemExit("from_pretrained must fail loudly without local weights")
except FileNotFoundError as e:
    assert "IA_PRETRAINED_DIR" in str(e)
# a local archive under $IA_PRETRAINED_DIR is found by the hub NAME
import os, torch
from indic_cl_asr_amd import checkpoint as ck
from indic_cl_asr_amd.config import model_config
src = M.EncDecHybridRNNTCTCModel(model_config("tiny", compute_dtype="fp32"))
os.environ["IA_PRETRAINED_DIR"] = %r
ck.write_nemo(src, os.path.join(%r, "ai4bharat__indicconformer_stt_hi_hybrid_rnnt_large.nemo"))
m = nemo_asr.models.ASRModel.from_pretrained("ai4bharat/indicconformer_stt_hi_hybrid_rnnt_large", strict=True,
                                             languages=src.cfg.languages, vocab_per_lang=src.cfg.vocab_per_lang, compute_dtype="fp32")
assert isinstance(m, M.EncDecHybridRNNTCTCModel)
a, b = src.state_dict(), m.state_dict()
assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
m.ctc_wer.log_prediction = False; m.wer.log_prediction = False                                   # R/cl_baseline.py:127-128
print("OK")
''' % (ROOT, str(tmp_path), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout, r.stderr)


def test_transcribe_config_fields_cover_what_the_scripts_pass():
    from indic_cl_asr_amd.model import InternalTranscribeConfig, TranscribeConfig
    names = {f.name for f in dataclasses.fields(TranscribeConfig)}
    assert {"batch_size", "return_hypotheses", "num_workers", "verbose", "logprobs", "language_id", "_internal"} <= names
    assert {"device", "temp_dir", "training_mode", "dither_value", "pad_to_value"} <= {f.name for f in dataclasses.fields(InternalTranscribeConfig)}
