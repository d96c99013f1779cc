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


def _bench(args, timeout=1500, **env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(FQH_BENCH_ONE_GPU="1", **env), cwd=ROOT,
                         capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_eight_ranks_hbm_resident_on_one_gpu():
    """bench.py --gpus 8 as eight gloo ranks on ONE GPU (the functional mode; an 8-GPU node is the driver's): cuts inside records at
    every rank, the device-side exchange in every step, the totals are the generator's; then one broken separator line in rank
    5's shard: every rank learns Parser::each's error — kind and record — from the MIN over the ranks (src/lib.rs:544-564)."""
    shard = 32 << 20
    j = _bench(["--gpus", "8", "--bytes", str(shard), "--steps", "2", "--warmup", "1", "--default-shard-stream-gib", "0.25", "--slot-mib", "32"],
               FQH_BENCH_SETTLE_STEPS="0")
    assert j["n_gpus"] == 8 and j["rccl"]["ranks"] == 8 and len(j["rccl"]["device_uuids"]) == 8
    assert j["rccl"]["via"].startswith("torch.distributed (gloo)")           # (RCCL does not put two ranks on one device)
    assert j["config"]["records_total"] == 8 * shard // 330 and shard % 330 != 0
    assert "0 of 3" in j["config"]["exchange"], j["config"]["exchange"]      # no fall-back to the host recipe on valid input
    # the default line carries the configs[4] leg: the same function at 8 ranks and, by rank 0 alone, at 1
    ss = j["sharded_stream"]
    blk = (32 << 20) // 2640 * 2640
    assert ss["ranks"] == 8 and ss["bytes_per_gpu"] == max(blk, (256 << 20) // blk * blk) + 997 and len(ss["numa"]) == 8
    assert len(ss["ring_setup_seconds"]) == 8 and all(x > 0 for x in ss["ring_setup_seconds"])
    for name in ("producer", "registered", "pinned_replay"):
        r = ss[name]
        assert r["check"]["phases_ok"] and r["check"]["histograms_ok"] and r["records"] == r["check"]["records_expected"]
        assert len(r["gbs_per_rank"]) == 8 and len(r["finish_seconds"]) == 8 and r["gbs_aggregate"] > 0
        assert r["n1_gbs"] > 0 and abs(r["ratio_vs_n1"] - r["gbs_aggregate"] / r["n1_gbs"]) < 0.01
    assert ss["producer"]["producer"]["threads_per_rank"] >= 1 and ss["pinned_replay"]["producer"]["threads_per_rank"] == 0
    assert ss["ratio_vs_n1"] == ss["producer"]["ratio_vs_n1"] and "world size 1" in ss["n1"]
    rec = (5 * shard + shard // 2) // 330
    j = _bench(["--gpus", "8", "--bytes", str(shard), "--steps", "1", "--warmup", "0", "--no-shard-stream"], FQH_BENCH_INJECT=str(rec * 330 + 177),
               FQH_BENCH_SETTLE_STEPS="0")
    assert j["n_gpus"] == 8 and j["first_error"]["status"] == 2 and j["first_error"]["n_records"] == rec, j["first_error"]
    # ... and an error injected into rank 5's range of the default line's streamed leg
    sshard = max(blk, (256 << 20) // blk * blk) + 997
    srec = (5 * sshard + sshard // 2) // 330
    j = _bench(["--gpus", "8", "--bytes", str(shard), "--steps", "1", "--warmup", "0", "--default-shard-stream-gib", "0.25", "--slot-mib", "32"],
               FQH_BENCH_INJECT_STREAM=str(srec * 330 + 177), FQH_BENCH_SETTLE_STEPS="0")
    fe = j["sharded_stream"]["first_error"]
    assert (fe["status"], fe["n_records"], fe["err_offset"], fe["key_rank"]) == (2, srec, srec * 330, 5), fe


@pytest.mark.gpu
def test_eight_ranks_sharded_and_streamed_on_one_gpu():
    """bench.py --gpus 8 --stream-gib G (configs[4]) as eight gloo ranks on one GPU: every rank streams its byte range phase-free,
    one exchange of ten words, every cut's record parsed by the rank it ends in; totals and histograms checked inside bench.py.
    Then a broken separator line in rank 5's range: the first-error key comes from rank 5 and names the oracle's record."""
    j = _bench(["--gpus", "8", "--stream-gib", "4", "--slot-mib", "32"])
    assert j["mode"] == "sharded-stream" and j["n_gpus"] == 8 and len(j["numa"]) == 8
    for name in ("producer", "registered", "pinned_replay"):
        assert j[name]["check"]["phases_ok"] and j[name]["check"]["histograms_ok"]
        assert j[name]["records"] == j[name]["check"]["records_expected"]
    blk = (32 << 20) // 2640 * 2640
    shard = max(blk, (4 << 30) // 8 // blk * blk) + 997
    assert j["bytes_per_gpu"] == shard
    rec = (5 * shard + shard // 2) // 330
    j = _bench(["--gpus", "8", "--stream-gib", "4", "--slot-mib", "32"], FQH_BENCH_INJECT=str(rec * 330 + 177))
    fe = j["first_error"]
    assert (fe["status"], fe["n_records"], fe["err_offset"], fe["key_rank"]) == (2, rec, rec * 330, 5), fe


@pytest.mark.gpu
def test_one_rank_through_the_library_rccl_binding():
    """bench.py --gpus 1 --comm abi end to end: the timed steps, one HBM-resident sharded step and the configs[4] leg with every
    gather / sum / min through fqh_comm_* / fqh_allgather / fqh_allreduce_u64 / fqh_allreduce_min_u64 (csrc/comm.hip; the gather
    of parallel_each's results, src/lib.rs:553-559) — a communicator of one rank, the most this box can give RCCL."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--comm", "abi", "--bytes", str(1 << 30), "--steps", "3",
                          "--warmup", "1", "--no-stats", "--no-stream", "--no-cpu-baseline", "--pmc-traffic", "off",
                          "--default-shard-stream-gib", "1", "--slot-mib", "32"], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 1 and j["rccl"]["via"].startswith("fqh_comm") and "protocol_check" in j["rccl"]
    ss = j["sharded_stream"]
    assert ss["comm"].startswith("fqh_comm") and ss["ranks"] == 1 and ss["ratio_vs_n1"] == 1.0 and ss["n1"] == "this run"
    for name in ("producer", "registered", "pinned_replay"):
        assert ss[name]["check"]["histograms_ok"] and ss[name]["records"] == ss[name]["check"]["records_expected"]
