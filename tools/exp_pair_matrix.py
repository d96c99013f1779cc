"""tools/exp_pair_matrix.py [LIB] — scan step time for every (context, input allocation) pair of one process: is the slow kind a
property of the input's allocation, of the context's workspace, or of the pair?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
cap = n // 300 + 16
NB, NC = int(os.environ.get("AB_INPUTS", "5")), int(os.environ.get("AB_CTXS", "4"))
L = C.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "fastq-rs_amd/libfastq_hip.so"))
L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
L.fqh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
rs = torch.empty(cap, dtype=torch.int64, device=dev)
bufs = [torch.empty(n + 4096, dtype=torch.uint8, device=dev) for _ in range(NB)]
ctxs = []
for j in range(NC):
    h = C.c_void_p(); assert L.fqh_create(0, C.byref(h)) == 0
    ctxs.append(h)
for b in bufs:
    assert L.fqh_synth_fill(ctxs[0], b.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
for rnd in range(2):
    print("round %d: rows = inputs, columns = contexts, ms per step (index kernel)" % rnd)
    for i, b in enumerate(bufs):
        row = []
        for h in ctxs:
            for _ in range(2): assert L.fqh_scan(h, b.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): L.fqh_scan(h, b.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
            t = (C.c_float * 5)(); L.fqh_last_timing(h, t)
            row.append("%.3f (%.3f)" % (dt, t[1]))
        print("  input %d: %s" % (i, "   ".join(row)), flush=True)
