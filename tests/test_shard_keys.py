"""The first-error exchange of the sharded, host-streamed mode on CPU, 2 ranks over gloo: every rank turns the gathered
words into the records it contributes and its packed (file offset, rank, kind) key (fqh_shard_stream_finish: host arithmetic
when this rank has no gap to parse), the MINIMUM over the ranks is the error Parser::each would return for the whole file —
what parallel_each returns when the parse fails (src/lib.rs:544-547, 561-564) — and fqh_shard_stream_outcome adds the per-rank
record slots up to the failing rank: the records delivered before the error, or all of them.  A rank that FAILS (its run, or
its finish) still takes part in every collective, and every rank learns of it."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
L = pkg.lib()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
OK, LEN = pkg.OK, pkg.E_LEN_MISMATCH
NW = pkg.SHARD_STREAM_WORDS
# words: status, n_records, n_newlines, phase, head_len, tail_len, err_offset, flags, lo, hi   (cuts on record boundaries: no gap)
cases = {{
    "clean":        [[OK, 10, 40, 0, 0, 0, 0, 0, 0, 1000], [OK, 7, 28, 0, 0, 0, 0, 0, 1000, 1700]],
    "error_rank1":  [[OK, 10, 40, 0, 0, 0, 0, 0, 0, 1000], [LEN, 3, 13, 0, 0, 0, 1300, 0, 1000, 1700]],
    "error_rank0":  [[LEN, 4, 17, 0, 0, 0, 400, 1, 0, 1000], [OK, 7, 28, 1, 0, 0, 0, 0, 1000, 1700]],   # rank 1 parsed under the phase its window gave: not its error
    "wrong_phase":  [[OK, 10, 41, 0, 0, 0, 0, 0, 0, 1000], [OK, 7, 28, 0, 0, 0, 0, 0, 1000, 1700]],     # 41 newlines in front of rank 1, which parsed at phase 0:
                                                                                                         # its records are void, it must parse [1000, 1700) itself — no GPU here: its finish fails
    "rank1_failed": [[OK, 10, 40, 0, 0, 0, 0, 0, 0, 1000], pkg.shard_failed_words(pkg.E_IO, 1000, 1700)],
    "empty_rank1":  [[OK, 10, 40, 0, 0, 0, 0, 0, 0, 1000], [OK, 0, 0, pkg.SHARD_EMPTY, 0, 0, 0, 0, 1000, 1000]],
    "rank1_unreadable": [[OK, 10, 40, 0, 0, 0, 0, 0, 0, 1000], [OK, 0, 0, pkg.SHARD_DEFER, 0, 0, 0, 3, 1000, 1700]],   # flags bit 1: its read callback failed — no failure
                                                                                                         # of the run: the last rank reads [1000, 1700) again in file order (no GPU here: that finish fails)
}}
want = {{"clean": (OK, 17, 0), "error_rank1": (LEN, 13, 1300), "error_rank0": (LEN, 4, 400), "wrong_phase": (pkg.E_DEVICE, 10, 1000),
        "rank1_failed": (pkg.E_IO, 10, 1000), "empty_rank1": (OK, 10, 0), "rank1_unreadable": (pkg.E_DEVICE, 10, 1000)}}
# ... and the words two ranks REALLY produced (fqh_shard_stream_run on an MI355X, tests/golden/shard_words_gpu.json: a file of
# 6000 records cut at a record start, so that no rank has a gap to parse and the finish is host arithmetic)
import json
fx = json.load(open(os.path.join({root!r}, "tests", "golden", "shard_words_gpu.json")))["stream_ranks2_cut_at_record_start"]
cases["gpu_words"] = fx["words"]
want["gpu_words"] = (fx["oracle"]["status"], fx["oracle"]["n_records"], 0)
flen = {{"gpu_words": fx["len"]}}
for name, rows in cases.items():
    mine = torch.tensor(rows[rank], dtype=torch.int64)
    allw = torch.zeros(world * NW, dtype=torch.int64)
    dist.all_gather_into_tensor(allw, mine)                         # the one exchange
    words = allw.numpy().astype(np.uint64)
    out = (C.c_uint64 * 2)()
    st = L.fqh_shard_stream_finish(None, pkg.READ_FN(), None, flen.get(name, 1700), words.ctypes.data, world, rank, 1 << 16, 2, 0, None, None, None, C.byref(out))
    if name in ("wrong_phase", "rank1_unreadable") and rank == 1:
        assert st == pkg.E_ARG, st                                  # (a gap to parse and no context: this rank's finish fails ...)
        out[0], out[1] = 0, pkg.shard_failure_key(rank, rows[rank][8], st)   # ... and it goes on with a failure key
    else:
        assert st == OK, (name, st)
    slots = torch.zeros(world, dtype=torch.int64)
    slots[rank] = int(out[0])
    key = torch.tensor([int(out[1]) - (1 << 63)], dtype=torch.int64)  # order-preserving map of the u64 key into i64
    dist.all_reduce(slots)
    dist.all_reduce(key, op=dist.ReduceOp.MIN)
    got = pkg.shard_stream_outcome(int(key.item()) + (1 << 63), slots.tolist())
    assert got == want[name], (name, rank, got, want[name])
dist.destroy_process_group()
open(os.path.join({out!r}, "ok_%d" % rank), "w").write("ok")
'''


def test_first_error_key_over_two_gloo_ranks(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()
