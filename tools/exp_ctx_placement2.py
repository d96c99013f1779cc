"""tools/exp_ctx_placement2.py PAD_MB [ORDER] — one context per process; PAD_MB of torch allocation between the 16 GiB input
(+ the offsets array) and the context's workspace.  ORDER=ws_first: the workspace is allocated BEFORE the input (a first scan of
a throw-away buffer of the same size that is freed again)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
pad_mb = int(sys.argv[1]); order = sys.argv[2] if len(sys.argv) > 2 else "input_first"
n = (16 << 30) // 330 * 330
cap = n // 300 + 16
L = C.CDLL(os.path.abspath(os.environ.get("FQH_LIB_PATH", "fastq-rs_amd/libfastq_hip.so")))
L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
L.fqh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
h = C.c_void_p(); assert L.fqh_create(0, C.byref(h)) == 0
if order == "ws_first":
    tmp = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
    L.fqh_synth_fill(h, tmp.data_ptr(), 0, n, 1)
    L.fqh_scan(h, tmp.data_ptr(), n, 1, None, None, 0, summ, carry)
    del tmp; torch.cuda.empty_cache()
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
rs = torch.empty(cap, dtype=torch.int64, device=dev)
pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
assert L.fqh_synth_fill(h, buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
for _ in range(3): assert L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
t = (C.c_float * 5)(); L.fqh_last_timing(h, t)
print("pad %5d MB, %s: %.3f ms per step (index %.3f emit %.3f)  buf %x rs %x" % (pad_mb, order, dt, t[1], t[3], buf.data_ptr(), rs.data_ptr()), flush=True)
