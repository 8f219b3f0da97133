"""bench.py end to end on the GPU box: the N = 1 line carries what the contract asks for, and `--gpus 2` re-launches itself
as two ranks (torch.distributed.run) -- on the one test GPU the ranks share cuda:0 over gloo (RCCL refuses two ranks on one
device), which is the code path the driver runs over RCCL on an 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_through_the_self_spawn_path():
    env = dict(os.environ, L2I_DIST_BACKEND="gloo", L2I_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 16 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
    assert d["env"]["L2I_DIST_BACKEND"] == "gloo"          # tuning / test switches are part of the line
    r = d["rank_ms_per_step"]                              # per-rank spread (a straggler among the ranks shows up here)
    assert 0 < r["min"] <= r["max"] <= d["ms_per_step"] * 1.001


def test_bench_refuses_wrong_result_switches():
    env = dict(os.environ, L2I_CONV_NOEPI="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)


def test_bench_default_line_carries_the_contract_fields():
    """`python bench.py` at N = 1 (few steps): ONE JSON line with the contract's keys, the roofline object (frac = achieved / peak,
    traffic from this round's PMC file, per-launch figures consistent with each other), the secondary legs (eager, f32_mode,
    generator forward with its precision modes: bf16x3 within the 1e-3 image bar of the exact-f32 mode), every timed step a replay."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["dtype"] == "bf16" and d["vs_baseline"] is None and "workload" in d["config"]
    assert "replay" in d["config"]["launch"]
    assert abs(d["value"] - 32 * 4 / (d["ms_per_step"] * 4e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["gflop_per_launch"] / r["avg_launch_us"] * 1e3) < 0.02 * r["achieved"]   # GFLOP / us = PFLOP/s = 1e3 TFLOP/s
    assert r["traffic"] is not None and r["traffic"] > r["algorithmic_bytes_per_launch"] and r["traffic_source"]["measured_in_run"] is False
    assert 0.15 < r["frac"] < 0.7 and 0.1 < r["wgrad_frac"] < 0.7
    assert d["eager"]["images_per_sec"] > 0.5 * d["value"] and d["f32_mode"]["images_per_sec"] > 0
    pm = d["g_forward"]["precision_modes"]
    assert pm["bf16x3"]["image_linf_vs_f32_mode"] < pm["bar"] < pm["bf16"]["image_linf_vs_f32_mode"]
    assert pm["bf16x3"]["ms"] < pm["f32"]["ms"]


def test_bench_reruns_itself_eagerly_when_graph_capture_fails():
    """A capture that is invalidated leaves the HIP runtime unusable for the rest of the process (tools/parity/capture_failure_probe.py),
    so bench.py's fallback is a fresh process with --no-graph: forced here by an illegal call inside the capture."""
    env = dict(os.environ, L2I_TEST_CAPTURE_FAIL="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
                        "--no-g-forward", "--no-f32-mode"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["config"]["launch"] == "eager" and d["value"] > 0 and "re-running eagerly" in p.stderr


def test_bench_line_accounts_for_the_collectives_of_a_data_parallel_step():
    """`comm` (N > 1, or the forced one-rank RCCL group): collectives per iteration, gradient bytes all-reduced (the flat buffers
    of G and D: 4 bytes x (40 853 873 + 62 696 387) parameters + padding), the stream time they held up, and the layer groups the
    gradient exchange is chunked into (arena.grad_groups; reference train_context_app_v2.py:108-110, model/sync_batchnorm/batchnorm.py:59-125)."""
    env = dict(os.environ, L2I_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-graph",
                        "--no-cpu-baseline", "--no-g-forward", "--no-f32-mode"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    c = _line(p.stdout)["comm"]
    assert c["collectives_per_step"] >= 20 and c["comm_exposed_ms"] >= 0.0
    assert 4 * (40853873 + 62696387) <= c["allreduce_bytes_per_step"] < 4.1 * (40853873 + 62696387)
    assert c["grad_groups"] == {"G": 4, "D": 4}
