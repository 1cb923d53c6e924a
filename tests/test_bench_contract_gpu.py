"""bench.py's output contract (the driver parses ONE JSON line from rank 0): keys, types and the internal consistency of the
line, on a short run of the headline configuration."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "6",
                        "--no-cpu-baseline", "--no-peaks"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], typ), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 6 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                     # BASELINE.md holds no published number for this metric
    assert d["unit"] == "utterances/s" and d["dtype"] == "bf16" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "BASELINE configs[1]" in cfg["workload"] and cfg["global_batch"] == 32 and cfg["parallelism"] == "dp1"
    assert "model" not in cfg
    # value = utterances of the whole job / time of the timed steps
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]
    # the timed steps are the reference-complete step: in-step decode + both WERs, monitor read after every optimizer step
    assert cfg["in_step_wer"]["enabled"] is True and 0.0 <= cfg["in_step_wer"]["training_batch_wer"]
    assert cfg["without_wer"]["in_step_wer"] is False and cfg["without_wer"]["ms_per_step"] > 0
    assert cfg["with_h2d_prefetch"]["h2d_bytes_per_step"] > 0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 2e-3 and 0.0 < rf["frac"] < 1.0
    assert "traffic" in rf
    assert "cpu_baseline" in d                          # (null here: --no-cpu-baseline; the default run fills it)
