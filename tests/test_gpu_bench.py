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


def test_bench_refuses_wrong_result_switches():
    env = dict(os.environ, L2I_CONV_NOEPI="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
