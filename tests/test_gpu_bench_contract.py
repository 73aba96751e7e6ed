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
    assert d["ranks_seen"] == 1
    # measured (not claimed) parity of the benched build against the reference-generated fixture, and the fp16 leg
    assert d["parity"]["mask_exact"] is True and d["parity"]["top1_equal"] is True and d["parity"]["logits_max_rel"] < 8e-3
    assert d["parity_fp16"]["logits_max_rel"] < 1e-3 and d["parity_fp16"]["mask_exact"] is True
    assert d["fp16"]["value"] > 50
    assert 0 < d["whole_path_frac_executed"] <= d["whole_path_frac_of_bf16_peak"]


def _run(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--warmup", "2", "--no-cpu-baseline", "--no-parity",
                        "--no-fp16-leg", "--streams", "2"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself_equal_one_rank():
    """`python bench.py --gpus 2` with no launcher starts its own two rank processes (gloo, both on cuda:0 of this 1-GPU
    box), really runs two ranks (ranks_seen), and the sharded accuracy accumulator over 2 x 6 items equals the one of a
    single rank over the same 12 items (item i -> rank i % world; what an item is does not depend on the world size)."""
    two = _run(["--gpus", "2", "--steps", "6", "--backend", "gloo", "--same-device"])
    one = _run(["--gpus", "1", "--steps", "12"])
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and one["ranks_seen"] == 1
    assert two["accuracy_accumulator"]["images"] == 12 == one["accuracy_accumulator"]["images"]
    for k in ("top1_hits", "top5_hits"):
        assert two["accuracy_accumulator"][k] == one["accuracy_accumulator"][k], k
