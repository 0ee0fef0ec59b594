"""bench.py --gpus N launches its own ranks.  On this ONE-GPU box the multi-process path is rehearsed
with QF_BENCH_REHEARSAL=1 (every rank on cuda:0, collectives over gloo): the JSON line must report the
process group that really ran, and without the rehearsal switch the launcher must refuse."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, rehearsal):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.pop("QF_BENCH_REHEARSAL", None)
    if rehearsal:
        env["QF_BENCH_REHEARSAL"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("extra", [["--batch", "4096", "--no-ring"], ["--workload", "sis"], ["--workload", "ringlpr"]])
def test_launcher_rehearsal_two_ranks(extra):
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spin-ms", "0", "--no-cpu"] + extra, rehearsal=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert len(out["devices"]) == 2 and "rehearsal" in out
    assert out["value"] > 0 and out["config"]["key_broadcast_ms"] > 0
    assert out["cpu_baseline"] is None


def test_launcher_refuses_two_ranks_on_one_gpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    p = _run(["--gpus", "2", "--workload", "sis", "--steps", "1"], rehearsal=False)
    assert p.returncode != 0 and p.stdout.strip() == "" and "refusing" in p.stderr
