"""tools/exp_pipeline.py — steps of the 16 GiB scan back to back: one blocking call per step (bench.py's step), against two contexts
that alternate (step k+1 is launched before step k is waited for), on one stream pair or on two."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
cap = n // 300 + 16
L = C.CDLL(os.path.abspath("fastq-rs_amd/libfastq_hip.so"))
vp = C.c_void_p
L.fqh_create.argtypes = [C.c_int, C.POINTER(vp)]
L.fqh_synth_fill.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64]
L.fqh_scan.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, vp, C.c_uint64, vp, vp]
L.fqh_scan_launch.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, vp, C.c_uint64]
L.fqh_scan_finish.argtypes = [vp, vp, vp]
L.fqh_set_stream.argtypes = [vp, vp]
L.fqh_last_timing.argtypes = [vp, vp]
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
def mk(stream=None):
    h = vp(); assert L.fqh_create(0, C.byref(h)) == 0
    if stream is not None: assert L.fqh_set_stream(h, vp(stream.cuda_stream)) == 0
    return h
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
A, B = mk(s1), mk(s2)
A2, B2 = mk(s1), mk(s1)
rsA = torch.empty(cap, dtype=torch.int64, device=dev); rsB = torch.empty(cap, dtype=torch.int64, device=dev)
assert L.fqh_synth_fill(A, buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
torch.cuda.synchronize()
def serial(h, rs, K):
    for _ in range(K):
        assert L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0
        assert summ[0] == n // 330
def piped(hs, K):
    pend = [False, False]
    for i in range(K):
        j = i & 1
        if pend[j]:
            assert L.fqh_scan_finish(hs[j][0], summ, carry) == 0 and summ[0] == n // 330
        assert L.fqh_scan_launch(hs[j][0], buf.data_ptr(), n, 1, None, hs[j][1].data_ptr(), cap) == 0
        pend[j] = True
    for j in range(2):
        if pend[j]:
            assert L.fqh_scan_finish(hs[j][0], summ, carry) == 0 and summ[0] == n // 330
for name, fn in (("serial, one context", lambda K: serial(A, rsA, K)),
                 ("two contexts, two streams", lambda K: piped([(A, rsA), (B, rsB)], K)),
                 ("two contexts, one stream", lambda K: piped([(A2, rsA), (B2, rsB)], K)),
                 ("serial, one context", lambda K: serial(A, rsA, K)),
                 ("two contexts, two streams", lambda K: piped([(A, rsA), (B, rsB)], K))):
    fn(6); torch.cuda.synchronize(); t0 = time.perf_counter(); K = 40
    fn(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
    t = (C.c_float * 5)(); L.fqh_last_timing(A, t)
    print("%-28s %.3f ms per step  %.0f GB/s  (context A's last step: index %.3f prefix %.3f emit %.3f)" % (name, dt, n / 1e6 / dt, t[1], t[2], t[3]), flush=True)
