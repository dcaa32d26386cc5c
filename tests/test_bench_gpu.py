"""bench.py as the driver runs it: `python bench.py --gpus N` must start N ranks itself (row (e) of SURVEY section 8).

On the one-GPU test box the two-rank case shares cuda:0 and aggregates over gloo (NDP_BENCH_BACKEND=gloo); the RCCL
branch itself is exercised with one rank under a launcher (NDP_BENCH_DIST=1: process group "nccl", barrier and the
all-reduce of the aggregate on the device).  The JSON lines are kept under gpurun_out/ (copied to profiles/ by hand).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (one warm-up step: on a fresh box the first step pays module loads and first launches, which made the rate comparisons below
#  fail once in a while)
SMALL = ["--steps", "2", "--warmup", "1", "--pairs-per-step", "48", "--slots", "24", "--engines", "1",
         "--no-cpu-baseline", "--no-roofline", "--no-latency", "--no-alt"]


def _run(cmd, env=None, timeout=1500):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                   # ONE line, from rank 0 only
    return json.loads(lines[0])


def _keep(name, rec):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump(rec, f)


@pytest.fixture(scope="module")
def one_rank():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return _run([sys.executable, "bench.py", "--gpus", "1"] + SMALL)


def test_bench_gpus_2_launches_two_ranks_itself(one_rank):
    rec = _run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, env={"NDP_BENCH_BACKEND": "gloo"})
    _keep("bench_gpus2_gloo.json", rec)
    assert rec["n_gpus"] == 2 and rec["config"]["backend"] == "gloo" and rec["scaling"] == "weak"
    # what the 8-GPU run will be judged by: the rank count, every rank's own rate and the cost of the one collective are in the line
    assert rec["world_size"] == 2 and len(rec["ranks"]["elapsed_s"]) == 2
    assert 0 < rec["ranks"]["pairs_per_s_min"] <= rec["ranks"]["pairs_per_s_max"] and rec["ranks"]["allreduce_ms"] >= 0
    assert rec["ranks"]["cpu_placement_rank0"]["cpus"] >= 1                      # each rank pinned to a CPU slice of its own
    assert abs(rec["value"] - 2 * 2 * 48 / max(rec["ranks"]["elapsed_s"])) < 1e-3 * rec["value"]   # sum of pairs / slowest rank
    # both ranks share ONE GPU here: twice the pairs in about twice the time, so the whole-job rate stays put
    assert 0.4 * one_rank["value"] < rec["value"] < 2.0 * one_rank["value"], (one_rank["value"], rec["value"])
    # weak scaling: every rank registered its own pairs (distinct seeds), the aggregate counts all of them
    assert abs(rec["ms_per_step"] * rec["value"] / 1e3 - 2 * 48) < 1e-6 * 96          # pairs per step over both ranks


def test_bench_rccl_branch_runs_on_one_rank(one_rank):
    rec = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "1"] + SMALL, env={"NDP_BENCH_DIST": "1"})
    _keep("bench_rccl_1rank.json", rec)
    assert rec["n_gpus"] == 1 and rec["config"]["backend"] == "rccl"
    assert 0.4 * one_rank["value"] < rec["value"] < 2.0 * one_rank["value"]
    assert rec["accuracy"].keys() == one_rank["accuracy"].keys()


def test_bench_line_carries_both_arithmetic_configurations():
    """The ONE line of the default run holds the headline measurement and, under `alt`, the same workload in the other
    arithmetic configuration (bitwise fp32-MFMA kernels <-> bf16-split kernels + matrix-pipe NN), each with the roofline of
    its own dominant kernel priced against the peak of the pipe it runs on."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    rec = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--pairs-per-step", "256", "--slots", "128",
                "--engines", "1", "--alt-steps", "1", "--no-cpu-baseline", "--no-latency"])
    _keep("bench_both_arith.json", rec)
    alt = rec["alt"]
    assert {rec["config"]["gemm_mode"], alt["gemm_mode"]} == {0, 7}
    for r in (rec, alt):
        split = r.get("gemm_mode", r.get("config", {}).get("gemm_mode")) == 7
        roof = r["roofline"]
        assert roof["unit"] == "TFLOP/s" and 0.05 < roof["frac"] < 1.0 and roof["achieved"] > 0
        if roof["kernel"].startswith(("k_eng_fwd8", "k_eng_bwd_f")):
            assert split and abs(roof["peak"] - 2500.0 / 3) < 1e-6       # three fp16 products per fp32-equivalent product
        else:
            assert roof["peak"] == 157.3
        assert r["value"] > 0 and r["dtype"] == ("f32 (fp16x2-split contractions, fp32 accumulate)" if split else "f32")
        assert set(r["kernels_ms_per_tick"]) >= ({"k_eng_fwd8", "k_eng_bwd_f"} if split else {"k_eng_fwd", "k_eng_bwd2", "k_eng_bwd1"})
        assert 0.02 < r["tick"]["frac"] < 1.0 and abs(r["tick"]["frac"] * r["tick"]["ms"] - r["tick"]["ideal_ms"]) < 1e-9
    # same workload, same early-stop behaviour: iterations per pair agree to a percent, accuracy to a few percent
    assert abs(rec["adam_iters_per_pair"] - alt["adam_iters_per_pair"]) < 0.02 * rec["adam_iters_per_pair"]
    assert abs(rec["accuracy"]["full-epe"] - alt["accuracy"]["full-epe"]) < 0.05 * rec["accuracy"]["full-epe"]
