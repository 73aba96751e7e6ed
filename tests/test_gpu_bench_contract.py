"""-m gpu: bench.py prints ONE JSON line with the keys the driver's contract names."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "images/sec"
    assert "workload" in d["config"] and "model" not in d["config"]
    r_ = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r_, k
    assert r_["bound"] == "mfma" and r_["unit"] == "TFLOP/s" and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-3
    assert d["value"] > 50 and abs(d["ms_per_step"] * d["value"] / 1e3 - 1.0) < 0.02     # value = n_gpus * steps / time
