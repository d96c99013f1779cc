"""The first-error exchange of the sharded, host-streamed mode on CPU, 2 ranks over gloo: every rank turns the gathered
words into its packed (global record, kind) key (fqh_shard_stream_finish: host arithmetic when no stitch has to be parsed),
the MINIMUM over the ranks is the error Parser::each would return for the whole file — what parallel_each returns when the
parse fails (src/lib.rs:544-547, 561-564) — and the SUM of the records is the count when there is none."""
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
# words: status, n_records, n_newlines, phase, head_len, tail_len, err_record, err_offset   (cuts on record boundaries: no stitch)
cases = {{
    "clean":        [[OK, 10, 40, 0, 0, 0, 0, 0], [OK, 7, 28, 0, 0, 0, 0, 0]],
    "error_rank1":  [[OK, 10, 40, 0, 0, 0, 0, 0], [LEN, 3, 13, 0, 0, 0, 3, 999]],
    "error_rank0":  [[LEN, 4, 17, 0, 0, 0, 4, 555], [OK, 7, 28, 1, 0, 0, 0, 0]],   # rank 1 parsed under the phase its window gave: not its error
    "wrong_phase":  [[OK, 10, 41, 0, 0, 0, 0, 0], [OK, 7, 28, 0, 0, 0, 0, 0]],     # 41 newlines in front of rank 1, which parsed at phase 0
}}
want = {{"clean": (OK, 17), "error_rank1": (LEN, 13), "error_rank0": (LEN, 4), "wrong_phase": (pkg.E_HEADER, 10)}}
for name, rows in cases.items():
    mine = torch.tensor(rows[rank], dtype=torch.int64)
    allw = torch.zeros(world * 8, dtype=torch.int64)
    dist.all_gather_into_tensor(allw, mine)                         # the one exchange (no tails here)
    words = allw.numpy().astype(np.uint64)
    out = (C.c_uint64 * 2)()
    st = L.fqh_shard_stream_finish(None, words.ctypes.data, None, 0, world, rank, None, 0, None, None, None, C.byref(out))
    assert st == OK, (name, st)
    rec = torch.tensor([int(out[0])], dtype=torch.int64)
    key = torch.tensor([int(out[1]) - (1 << 63)], dtype=torch.int64)  # order-preserving map of the u64 key into i64
    dist.all_reduce(rec)
    dist.all_reduce(key, op=dist.ReduceOp.MIN)
    status, err_record = pkg.error_key_unpack(int(key.item()) + (1 << 63))
    got = (status, int(rec.item()) if status == OK else err_record)
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
