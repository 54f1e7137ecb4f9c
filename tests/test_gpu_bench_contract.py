"""GPU: bench.py's one-line JSON contract (the driver parses it), on a reduced geometry so the test stays short."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--height", "128", "--width", "256", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "alt_conv_math", "other_workloads"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["unit"] == "images/s" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 2 * 1e3 / j["ms_per_step"]) < 1e-2 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("mfma", "hbm") and r["unit"] == ("TFLOP/s" if r["bound"] == "mfma" else "GB/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["mfma_view"]["unit"] == "TFLOP/s" and r["hbm_view"]["unit"] == "GB/s" and "by_instantiation" in r
    assert r["traffic"] is None or r["traffic_source"]  # a traffic figure always names the PMC file it came from
    assert j["cpu_baseline"] is None  # --no-cpu-baseline
    alt = j["alt_conv_math"]
    assert alt["conv_math"] in ("f32", "bf16x3") and alt["conv_math"] != r["conv_math"] and alt["value"] > 0
    ow = j["other_workloads"]  # configs[4] arithmetic always; configs[2] only from 256 x 256 up (critics)
    assert ow["seg_bf16"]["dtype"] == "bf16" and ow["seg_bf16"]["value"] > 0 and "confmax_seg" not in ow


def test_bench_self_launches_its_workers():
    """`python bench.py --gpus 2` from a bare shell (no torchrun environment) must start its own workers before touching
    the GPU, print rank 0's line and exit 0.  Rehearsed on the one GPU: both ranks share it and talk over gloo."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HN_BENCH_SHARE_GPU="1", HN_BENCH_BACKEND="gloo", HEATNET_AUTOTUNE="0")  # (no per-shape candidate timing: launch logic is the subject)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "1", "--height", "256", "--width", "256", "--workload", "confmax_seg"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and j["scaling"] == "weak"


def test_bench_rccl_rehearsal_on_one_rank():
    """HN_BENCH_FORCE_DIST=1: the N=1 run takes the N>1 code path on the `nccl` (RCCL) backend -- process group, parameter
    broadcast, bucketed all-reduces from the backward hooks, barriers, max-over-ranks -- and still prints the contract line."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HN_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "1",
                          "--height", "256", "--width", "256", "--workload", "confmax_seg", "--no-cpu-baseline", "--no-alt"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0
    assert j["rehearsal"]["backend"] == "nccl" and j["rehearsal"]["collectives_issued"] > 0


def test_bench_worker_failure_is_reported():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HN_BENCH_SHARE_GPU="1", HN_BENCH_BACKEND="gloo", HN_BENCH_FAIL_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch", "2", "--height", "128", "--width", "256", "--no-roofline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0


def test_bench_bf16_workload_line():
    """configs[4] arithmetic as its own workload: dtype "bf16", never mixed into the f32 headline."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--height", "128", "--width", "256", "--no-cpu-baseline", "--workload", "seg_bf16"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert j["dtype"] == "bf16" and "configs[4]" in j["config"]["workload"] and j["alt_conv_math"] is None
    assert j["metric"] == "training images/s at 128x256 RGB+IR, bs=2/GPU"
    assert j["roofline"]["conv_math"] == "bf16" and j["roofline"]["mfma_view"]["peak"] == 2500.0
