"""-m gpu: bench.py prints ONE JSON line with the keys the driver's contract names."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--sustain-seconds", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
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
    # ---- ONE protocol for both operand builds; the headline is the build whose MEASURED parity meets the north_star's
    # selection-mask + logit tolerance (fp16 operands = the reference's autocast dtype, ttl.py:79), named as such
    assert set(d["legs"]) == {"fp16", "bf16"} and d["protocol"]["legs"] == ["fp16", "bf16"] and d["protocol"]["interleaved_blocks"] is True
    assert d["dtype"] == "fp16" and d["dtype_conforming"] == "fp16" and d["value_conforming"] == d["value"] and "ttl.py:79" in d["dtype_note"]
    f16, b16 = d["legs"]["fp16"], d["legs"]["bf16"]
    assert f16["is_headline"] is True and b16["is_headline"] is False and f16["value"] == d["value"]
    for leg in (f16, b16):      # same steps, warm-up, repeats and graph regime for every leg
        assert (leg["steps"], leg["warmup"], leg["repeats"], leg["hip_graph"]) == (6, 2, d["repeats"], d["config"]["hip_graph"])
        assert leg["value"] > 50 and leg["value_min"] <= leg["value"] <= leg["value_max"] and leg["value_iqr"] >= 0
        assert leg["lib_path"].endswith(".so") and len(leg["lib_sha256_16"]) == 16
        rf = leg["roofline"]
        assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 100
        assert set(leg["parity"]["meets_north_star_tolerance"]) == {"selection_mask", "logits", "lora_weights", "lora_weights_frobenius",
                                                                    "lora_gradients", "all"}
        assert leg["parity"]["mask_exact"] is True and leg["parity"]["top1_equal"] is True
        for k in ("lora_weights_rel_frobenius", "lora_update_rel_frobenius", "grad_rel_frobenius"):
            assert 0 <= leg["parity"][k] < 1.0, k
    assert f16["lib_path"].endswith("libttl_hip_fp16.so") and b16["lib_path"].endswith("libttl_hip.so") and d["lib_path"] == f16["lib_path"]
    assert {k: v for k, v in d["parity"].items() if k != "strict"} == f16["parity"]
    # the test-only fp32 build on the same fixture, held to the tolerance by the letter (never timed: no leg of its own)
    st = d["parity"]["strict"]
    assert st["dtype"] == "strict" and "strict" not in d["legs"]
    assert st["meets_north_star_tolerance"] == {"selection_mask": True, "logits": True, "lora_gradients": True, "lora_weights": True, "all": True}, st
    assert st["logits_max_rel"] <= 1e-5 and st["grad_max_rel"] <= 1e-4 and st["lora_weights_elements_beyond_tolerance_not_exempt"] == 0
    assert d["value_fp16"] == f16["value"] and d["value_bf16"] == b16["value"] and d["headline_conforms"] is True
    assert f16["parity"]["meets_north_star_tolerance"]["logits"] is True and f16["parity"]["meets_north_star_tolerance"]["selection_mask"] is True
    assert f16["parity"]["logits_max_rel"] <= 1e-3 and f16["parity"]["adapted_logits_max_rel"] <= 1e-3
    assert b16["parity"]["logits_max_rel"] < 6e-3 and b16["parity"]["meets_north_star_tolerance"]["logits"] is False
    # the conforming build costs at most a few per cent (same kernels, same MFMA rate; the chip holds a lower clock on fp16 operands)
    assert f16["value"] > 0.93 * b16["value"]
    assert 0 < d["whole_path_frac_executed"] <= d["whole_path_frac_of_bf16_peak"]
    # the headline is the MEDIAN of `repeats` timed blocks of `steps` steps each; short blocks (6 x 3.5 ms) are repeated until
    # they cover 1.5 s (at most 25); the spread and the host's share are in the line
    assert 10 <= d["repeats"] <= 25 and d["value_min"] <= d["value"] <= d["value_max"] and d["value_iqr"] >= 0
    assert 0 < d["host_enqueue_ms_per_image"] < d["ms_per_step"] * 1.05
    assert isinstance(d["config"]["hip_graph"], bool) and d["config"]["lora_targets"] == ["q_proj", "v_proj"]
    # conformance is decided over every deciding fixture (named in headline_rule), the others are listed with their verdicts
    par = d["parity"]
    assert par["conforms_on_every_deciding_fixture"] is True and par["fixtures_deciding"][0] == "b16_n64_k200_ent0" and len(par["fixtures_deciding"]) >= 6
    assert all(par["fixtures"][fx]["selection_mask_and_logits_within_tolerance"] for fx in par["fixtures_deciding"])
    assert all(fx in d["headline_rule"] for fx in par["fixtures_deciding"]) and "b16_n64_k200_qkvo" in par["fixtures"]
    assert b16["parity"]["conforms_on_every_deciding_fixture"] is False
    assert st["conforms_on_every_deciding_fixture"] is True and st["fixtures_outside_tolerance"] == []
    # the frozen kernel path: the five run-time switches, all at their defaults, are in the line
    ke = d["protocol"]["kernel_env"]
    assert set(ke) == {"TTL_GEMM_HUGE", "TTL_GEMM_HUGE_NARROW", "TTL_GEMM_HUGE_MIN_FILL", "TTL_BWD_COMPACT", "TTL_CONCURRENCY"}
    assert d["protocol"]["kernel_env_all_default"] is True and all(v["value"] == v["default"] for v in ke.values())
    # a continuous run of the headline leg (no fence between images) beside the 1.5-s timed blocks
    su = d["sustained"]
    assert su["dtype"] == "fp16" and 3.5 < su["seconds"] < 15 and su["images"] >= 6 * 50 and su["value"] > 50
    assert 0.85 < su["ratio_to_value"] < 1.15 and su["value_last_20s"] > 50
    ts = d["torch_stack_same_gpu"]
    assert ts["static"] is True and ts["value"] > 10 and ts["source"].startswith("profiles/") and ts["ratio"] > 2


def test_bench_refuses_a_kernel_switch_off_its_default_unless_asked():
    """A stray TTL_GEMM_HUGE=0 (or any other run-time switch of the library off its default) in the environment would change the
    measured kernel path silently: refused, like a swapped library; --variant-env allows it and the line records it."""
    env = dict(os.environ, TTL_GEMM_HUGE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--variant-env" in (r.stderr + r.stdout) and "TTL_GEMM_HUGE=0" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    # a closed experiment's name is not a switch of the product build: it changes nothing and is not refused
    d = _run(["--steps", "6", "--repeats", "2", "--precision", "fp16", "--sustain-seconds", "0"], env=dict(os.environ, TTL_ATTN_VARIANT="1", TTL_GEMM_BIG="0"))
    assert d["protocol"]["kernel_env_all_default"] is True
    d = _run(["--steps", "6", "--repeats", "2", "--precision", "fp16", "--sustain-seconds", "0", "--variant-env"], env=env)
    assert d["protocol"]["kernel_env"]["TTL_GEMM_HUGE"] == {"value": 0, "default": 2} and d["protocol"]["kernel_env_all_default"] is False


def test_bench_refuses_a_swapped_library_unless_asked():
    """TTL_HIP_LIB_* overrides (the A/B tools' way of loading another build) are an error for a plain bench run."""
    env = dict(os.environ, TTL_HIP_LIB_BF16=os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "ttl_amd", "libttl_hip.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--variant-lib" in (r.stderr + r.stdout)
    d = _run(["--steps", "6", "--repeats", "2", "--variant-lib"], env=env)
    assert d["protocol"]["variant_lib_env"] == ["TTL_HIP_LIB_BF16"] and d["lib_path"].endswith("libttl_hip.so")


def test_bench_runs_the_north_star_adapter_set():
    """--lora-targets qkvo: adapters on q, k, v and out_proj (BASELINE.json north_star; the reference ships q, v)."""
    d = _run(["--steps", "6", "--repeats", "2", "--lora-targets", "qkvo"])
    assert d["config"]["lora_targets"] == ["q_proj", "k_proj", "v_proj", "out_proj"] and "q+k+v+out" in d["config"]["workload"]
    assert d["value"] > 50 and d["repeats"] >= 2 and d["dtype"] == "bf16" and list(d["legs"]) == ["bf16"]


def test_rccl_process_group_of_one_rank_carries_the_accumulator():
    """backend "nccl" IS RCCL on ROCm: a world of ONE rank on this box initialises it, and ImageShard's all-reduce of a DEVICE
    tensor goes through it (catches RCCL / HSA IPC environment problems on a lease before the first multi-GPU run does)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "ttl-test-time-low-rank-adaptation_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from ttl_amd.driver import ImageShard
class Two(ImageShard):                       # world 1 short-cuts the collective: force it through the backend
    def _reduce(self, t, op):
        assert t.is_cuda and dist.get_backend() == "nccl"
        dist.all_reduce(t, op=op)
        return t
s = Two(0, 1)
acc = s.accuracy(torch.tensor([3, 4, 5], dtype=torch.int64, device="cuda"))
assert (acc["hits1"], acc["hits5"], acc["count"]) == (3, 4, 5), acc
assert s.ranks_seen("cuda") == 1
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
print("RCCL_OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _run(extra, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--warmup", "2", "--no-cpu-baseline", "--no-parity",
                        "--precision", "bf16", "--streams", "2", "--sustain-seconds", "0"] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
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
    # the N > 1 line shows every rank: its own rate before the closing barrier, the balance verdict, the graph regime per rank
    assert len(two["per_rank_value"]) == 2 and all(v > 10 for v in two["per_rank_value"])
    assert set(two["rank_balance"]) >= {"slowest_over_median", "ok"} and two["config"]["hip_graph_per_rank"] == [True, True]
    assert two["config"]["backend"] == "gloo" and "per_rank_value" not in one
