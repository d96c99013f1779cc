"""bench.py --gpus N launches itself (with WORLD_SIZE unset it used to benchmark ONE GPU and print n_gpus: 1).
CPU: on a node with fewer GPUs than asked for it refuses, loudly.  GPU: two ranks on one GPU (functional mode, gloo) print
a line with n_gpus == 2 and the ranks' device ids.  The gather it times is the analogue of parallel_each's,
src/lib.rs:521-559."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FQH_BENCH_ONE_GPU", "FQH_BENCH_BACKEND")}
    env.update(kw)
    return env


def test_gpus_n_without_enough_gpus_refuses():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs: the launch would go through")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], "no benchmark line may be printed"
    assert "--gpus 2" in out.stderr


def test_world_size_mismatch_refuses():
    """Under a launcher with the wrong number of ranks the line would lie about n_gpus: refuse before any GPU work."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                         env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
                                  FQH_BENCH_BACKEND="gloo", FQH_BENCH_ONE_GPU="1"),
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_gpus_2_launches_itself_on_one_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bytes", str(1 << 30), "--steps", "3",
                          "--warmup", "1"], env=_env(FQH_BENCH_ONE_GPU="1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl"]["ranks"] == 2 and len(j["rccl"]["device_uuids"]) == 2
    assert j["config"]["records_total"] == 2 * (1 << 30) // 330
    assert "0 of 4" in j["config"]["exchange"], j["config"]["exchange"]   # the device-side exchange was used in every step
