"""tools/exp_lines_offset.py — VERDICT r4 item 5: is there an aliasing stride between the input's allocation and the fast path's
line buffer that makes a pair "slow"?  (tools/bin/tune.so: a -DFQH_TUNING build, tools/build_tuning.sh.)
1. the pair matrix (inputs x contexts, adaptation off) finds a slow and a fast pair;
2. the LINE BUFFER of each is shifted inside its allocation (FQH_TUNE_LINES_OFFSET: 128 B .. 3 MiB) — index kernel time per offset;
3. the INPUT of the slow pair is shifted inside its allocation (16 B .. 3 MiB; the bytes are generated again at each place).
Virtual addresses are printed with every row.  Index kernel time = HIP events of the real fqh_scan calls (fqh_last_timing)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FQH_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "tune.so"))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
GIB = float(os.environ.get("EXP_GIB", "16"))
n = int(GIB * (1 << 30)) // 330 * 330
cap = n // 300 + 16
SLACK = 4 << 20
NB, NC = int(os.environ.get("EXP_INPUTS", "4")), int(os.environ.get("EXP_CTXS", "3"))
rs = torch.empty(cap, dtype=torch.int64, device=dev)
stores = [torch.empty(n + SLACK + 4096, dtype=torch.uint8, device=dev) for _ in range(NB)]
ctxs = []
for j in range(NC):
    c = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    c.set_adapt_lines(0)
    c.set_spin_wait(20000)
    ctxs.append(c)
for b in stores:
    ctxs[0].synth_fill(b.data_ptr(), 0, n)


def t_index(ctx, ptr, reps=5, warm=2):
    best = 1e9
    for i in range(warm + reps):
        s = ctx.scan(ptr, n, True, None, rs.data_ptr(), cap)[0]
        assert s.n_records == n // 330 and ctx.last_scan_fast()
        if i >= warm:
            best = min(best, ctx.timing().index_ms)
    return best


os.environ["FQH_DEBUG_WS"] = "1"   # (the library prints its workspace addresses when it allocates them)
print("== pair matrix: rows = inputs, columns = contexts; index kernel ms (min of 5)", flush=True)
mat = {}
for i, b in enumerate(stores):
    row = []
    for j, c in enumerate(ctxs):
        mat[(i, j)] = t_index(c, b.data_ptr())
        row.append("%.3f" % mat[(i, j)])
    print("  input %d @ %#x: %s" % (i, b.data_ptr(), "  ".join(row)), flush=True)
os.environ.pop("FQH_DEBUG_WS")
slow = max(mat, key=mat.get)
fast = min(mat, key=mat.get)
print("slowest pair (input %d, ctx %d) %.3f ms; fastest (input %d, ctx %d) %.3f ms; spread %.1f %%" % (
    slow[0], slow[1], mat[slow], fast[0], fast[1], mat[fast], 100 * (mat[slow] / mat[fast] - 1)), flush=True)
OFFS = [0, 128, 256, 384, 512, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 98304, 131072, 262144, 524288, 786432,
        1 << 20, (1 << 20) + 4096, (1 << 20) + 65536, 3 << 19, 2 << 20, (2 << 20) + 4096, 3 << 20, 0]
for name, (i, j) in (("slow", slow), ("fast", fast)):
    print("== line buffer shifted inside its allocation, %s pair (input %d, ctx %d)" % (name, i, j), flush=True)
    for off in OFFS:
        os.environ["FQH_TUNE_LINES_OFFSET"] = str(off)
        print("  lines +%8d B: %.3f ms" % (off, t_index(ctxs[j], stores[i].data_ptr(), reps=4)), flush=True)
    os.environ.pop("FQH_TUNE_LINES_OFFSET")
i, j = slow
print("== input shifted inside its allocation, slow pair (input %d @ %#x, ctx %d)" % (i, stores[i].data_ptr(), j), flush=True)
for off in [0, 16, 256, 4096, 8192, 16384, 65536, 262144, 1 << 20, 3 << 19, 2 << 20, 3 << 20, 0]:
    ctxs[0].synth_fill(stores[i].data_ptr() + off, 0, n)
    torch.cuda.synchronize()
    print("  input +%8d B: %.3f ms" % (off, t_index(ctxs[j], stores[i].data_ptr() + off, reps=4)), flush=True)
# ... and the other way round: does a FAST pair turn slow anywhere?
i, j = fast
print("== input shifted inside its allocation, fast pair (input %d @ %#x, ctx %d)" % (i, stores[i].data_ptr(), j), flush=True)
for off in [0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 0]:
    ctxs[0].synth_fill(stores[i].data_ptr() + off, 0, n)
    torch.cuda.synchronize()
    print("  input +%8d B: %.3f ms" % (off, t_index(ctxs[j], stores[i].data_ptr() + off, reps=4)), flush=True)
