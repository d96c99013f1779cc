"""ctypes binding of oracle/libfqref.so — TEST INFRASTRUCTURE ONLY (see oracle/fqref.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BUFSIZE = 68 * 1024
SYNTH_RECLEN = 330
SYNTH_SEED = 0x5EEDF00D2026
NSCALARS = 8

OK, E_HEADER, E_SEP, E_LEN, E_TRUNCATED, E_TOO_LONG = range(6)


class Idx(C.Structure):
    _fields_ = [("start", C.c_uint64), ("head", C.c_uint64), ("seq", C.c_uint64),
                ("sep", C.c_uint64), ("qual", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("stopped", C.c_int32), ("n_records", C.c_uint64),
                ("bytes_consumed", C.c_uint64)]


def build():
    """(Re)build libfqref.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfqref.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u8p, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.fqref_count.argtypes = [u8p, u64, u64, u64, C.POINTER(Result)]
        L.fqref_count_file.argtypes = [C.c_char_p, u64, C.POINTER(Result)]
        L.fqref_each_zipped.argtypes = [u8p, u64, u8p, u64, u64, u8p, u64, C.c_void_p, u64, C.POINTER(u64),
                                        C.POINTER(C.c_int32 * 2), C.POINTER(C.c_int32)]
        L.fqref_index.argtypes = [u8p, u64, u64, u64, C.c_void_p, u64, C.POINTER(Result)]
        L.fqref_offsets.argtypes = [u8p, u64, u64, u64, C.c_void_p, u64, C.POINTER(Result)]
        L.fqref_stats.argtypes = [u8p, u64, u64, u64, u32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(Result)]
        L.fqref_record_sets.argtypes = [u8p, u64, u64, u64, u32, C.c_void_p, u64,
                                        C.POINTER(u64), C.c_void_p, C.POINTER(Result)]
        L.fqref_synth_range.argtypes = [C.c_void_p, u64, u64, u64]
        L.fqref_strerror.restype = C.c_char_p
        L.fqref_strerror.argtypes = [C.c_int]
        L.fqref_validate_dna.argtypes = [u8p, u64]
        L.fqref_validate_dnan.argtypes = [u8p, u64]
        _LIB = L
    return _LIB


def _buf(data):
    """bytes / bytearray / np.uint8 array -> (keepalive, pointer, length)."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(data), dtype=np.uint8)
    return a, a.ctypes.data if a.size else None, a.size


def count(data, bufsize=BUFSIZE, max_read=0):
    a, p, n = _buf(data)
    r = Result()
    lib().fqref_count(p, n, bufsize, max_read, C.byref(r))
    return r


def count_file(path, bufsize=BUFSIZE):
    """examples/fastq-count.rs on a plain file: the parser reads it through its 68 KiB Buffer, one read(2) per refill."""
    r = Result()
    if lib().fqref_count_file(os.fsencode(path), bufsize, C.byref(r)) != 0:
        raise OSError("cannot open %s" % path)
    return r


def each_zipped(data1, data2, flags, bufsize=BUFSIZE):
    """each_zipped (src/lib.rs:577-609) with the callback scripted by `flags` (bytes, 2 bits each, cycled).
    -> (status, (finished1, finished2), trace[ncalls, 2] of record indices, UINT64_MAX = None)."""
    a1, p1, n1 = _buf(data1)
    a2, p2, n2 = _buf(data2)
    af, pf, nf = _buf(flags)
    cap = n1 // 6 + n2 // 6 + 8
    trace = np.zeros((cap, 2), dtype=np.uint64)
    ncalls, fin, st = C.c_uint64(0), (C.c_int32 * 2)(), C.c_int32(0)
    lib().fqref_each_zipped(p1, n1, p2, n2, bufsize, pf, nf, trace.ctypes.data, cap, C.byref(ncalls), C.byref(fin), C.byref(st))
    return st.value, (bool(fin[0]), bool(fin[1])), trace[: ncalls.value]


def index(data, bufsize=BUFSIZE, max_read=0):
    """-> (Result, np structured array of (start, head, seq, sep, qual))."""
    a, p, n = _buf(data)
    r = Result()
    cap = max(1, n // 6 + 1)  # the shortest legal record "@\n\n+\n\n" has 6 bytes
    out = np.zeros((cap, 5), dtype=np.uint64)
    lib().fqref_index(p, n, bufsize, max_read, out.ctypes.data, cap, C.byref(r))
    return r, out[: r.n_records]


def offsets(data, bufsize=BUFSIZE, max_read=0):
    a, p, n = _buf(data)
    r = Result()
    cap = max(1, n // 6 + 1)
    out = np.zeros(cap, dtype=np.uint64)
    lib().fqref_offsets(p, n, bufsize, max_read, out.ctypes.data, cap, C.byref(r))
    return r, out[: r.n_records]


def stats(data, lmax, bufsize=BUFSIZE, max_read=0):
    """-> (Result, qual_hist[lmax,256], base_hist[lmax,8], scalars[8]) as uint64 arrays."""
    a, p, n = _buf(data)
    r = Result()
    qh = np.zeros((lmax, 256), dtype=np.uint64)
    bh = np.zeros((lmax, 8), dtype=np.uint64)
    sc = np.zeros(NSCALARS, dtype=np.uint64)
    lib().fqref_stats(p, n, bufsize, max_read, lmax, qh.ctypes.data, bh.ctypes.data,
                      sc.ctypes.data, C.byref(r))
    return r, qh, bh, sc


def record_sets(data, n_threads=1, bufsize=BUFSIZE, max_read=0):
    """-> (Result, set_sizes, worker_counts)."""
    a, p, n = _buf(data)
    r = Result()
    cap = max(4, 2 * (n // max(1, min(bufsize // 2, max_read or bufsize))) + 8)   # (one set per refill: per read of a reader that comes back short)
    sizes = np.zeros(cap, dtype=np.uint64)
    workers = np.zeros(max(1, n_threads), dtype=np.uint64)
    nsets = C.c_uint64(0)
    lib().fqref_record_sets(p, n, bufsize, max_read, n_threads, sizes.ctypes.data, cap,
                            C.byref(nsets), workers.ctypes.data, C.byref(r))
    return r, sizes[: min(cap, nsets.value)], workers


def accessors(data, idx_row):
    """head(), seq(), qual() of one record as bytes (records.rs:75-90)."""
    start, head, seq, sep, qual = (int(x) for x in idx_row)
    d = bytes(data[start: start + qual + 1]) if not isinstance(data, (bytes, bytearray)) \
        else data[start: start + qual + 1]

    def trim(b):
        return b[:-1] if b.endswith(b"\r") else b

    return trim(d[1:head]), trim(d[head + 1: seq]), trim(d[sep + 1: qual])


def synth(byte_off, length, seed=SYNTH_SEED):
    out = np.empty(length, dtype=np.uint8)
    lib().fqref_synth_range(out.ctypes.data, byte_off, length, seed)
    return out


def strerror(status):
    return lib().fqref_strerror(status).decode()
