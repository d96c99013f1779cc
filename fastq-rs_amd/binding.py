"""ctypes stub over libfastq_hip.so (include/fastq_hip.h).  Plumbing only: device memory comes from
the caller (torch tensors -> data_ptr(), or fqh_dev_alloc); every call goes through the C ABI."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FQH_LIB_PATH: a tuning build of the same library, tools/exp_fztime.sh; never a fallback)
LIB_PATH = os.environ.get("FQH_LIB_PATH") or os.path.join(_HERE, "libfastq_hip.so")

__all__ = ["LIB_PATH", "lib", "Ctx", "Stream", "Comm", "COMM_ID_BYTES", "Chunk", "ShardResult", "READ_FN", "SHARD_STREAM_WORDS", "NO_ERROR_KEY", "shard_stream_outcome", "shard_failed_words", "shard_failure_key", "SHARD_EMPTY", "SHARD_PASS", "SHARD_DEFER", "STREAM_INDEX", "STREAM_STATS", "STREAM_TIMING", "StreamTimes", "Carry", "Summary", "Timing", "IdxRecord", "FqhError",
           "strerror", "carry_combine", "OK", "E_HEADER", "E_SEP", "E_LEN_MISMATCH", "E_TRUNCATED", "E_TOO_LONG",
           "E_IO", "E_DEVICE", "E_ARG", "E_CAPACITY", "E_AGAIN", "SHARD_WORDS", "BUFSIZE", "NSCALARS", "OPT_FAST_PATH", "OPT_SINGLE_PASS", "OPT_PLACE_TRIES", "OPT_SPIN_WAIT", "OPT_REUSE_INDEX", "OPT_ADAPT_LINES", "OPT_OWN_STREAM_NONBLOCKING", "OPT_KEEP_RING", "STREAM_EXTERNAL", "MAP_FN", "EXPORTS"]

OK, E_HEADER, E_SEP, E_LEN_MISMATCH, E_TRUNCATED, E_TOO_LONG, E_IO, E_DEVICE, E_ARG, E_CAPACITY, E_AGAIN = range(11)
SHARD_WORDS = 8
BUFSIZE = 68 * 1024
NSCALARS = 8
OPT_FAST_PATH, OPT_SINGLE_PASS, OPT_PLACE_TRIES, OPT_SPIN_WAIT, OPT_REUSE_INDEX, OPT_ADAPT_LINES, OPT_OWN_STREAM_NONBLOCKING, OPT_KEEP_RING = 1, 2, 3, 4, 5, 6, 7, 8

# every symbol include/fastq_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "fqh_create", "fqh_destroy", "fqh_strerror", "fqh_last_error", "fqh_abi_version",
    "fqh_set_stream", "fqh_set_bufsize", "fqh_set_option", "fqh_last_scan_fast", "fqh_last_stats_route", "fqh_placement", "fqh_line_buffers", "fqh_scan", "fqh_scan_launch", "fqh_scan_finish",
    "fqh_shard_prescan", "fqh_shard_prescan_launch", "fqh_shard_rescan_launch", "fqh_shard_align", "fqh_stream_carry", "fqh_carry_combine", "fqh_rescan_launch", "fqh_invalidate", "fqh_index_records", "fqh_record_flags", "fqh_gather_records", "fqh_len_hist", "fqh_stats", "fqh_stats_launch", "fqh_stats_finish", "fqh_stats_launch_lead",
    "fqh_scan_stats", "fqh_scan_stats_launch", "fqh_scan_stats_finish", "fqh_last_timing",
    "fqh_stream_create", "fqh_stream_destroy", "fqh_stream_set_stats", "fqh_stream_acquire", "fqh_stream_submit",
    "fqh_stream_collect", "fqh_stream_release", "fqh_stream_timing", "fqh_stream_note_read", "fqh_comm_unique_id", "fqh_comm_create", "fqh_comm_destroy", "fqh_allgather",
    "fqh_allreduce_u64", "fqh_allreduce_min_u64", "fqh_sync", "fqh_shard_stream_run", "fqh_shard_result_words",
    "fqh_shard_failed_words", "fqh_shard_failure_key",
    "fqh_shard_stream_finish", "fqh_shard_stream_outcome", "fqh_stream_set_origin", "fqh_synth_fill", "fqh_read_ceiling", "fqh_dev_alloc", "fqh_dev_free", "fqh_memcpy_h2d",
    "fqh_memcpy_d2h", "fqh_memset", "fqh_stream_submit_external", "fqh_host_register", "fqh_host_unregister",
    "fqh_shard_stream_run_mapped", "fqh_stream_release_chunk",
]


class Carry(C.Structure):
    _fields_ = [("base_offset", C.c_uint64), ("nl_count", C.c_uint64), ("back", C.c_uint64 * 4)]


class Summary(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("bytes_consumed", C.c_uint64),
                ("parse_status", C.c_int32), ("reserved", C.c_int32),
                ("err_record", C.c_uint64), ("err_offset", C.c_uint64),
                ("n_newlines", C.c_uint64), ("tail_len", C.c_uint64),
                ("max_record_len", C.c_uint64), ("n_line_starts", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("index_ms", C.c_float), ("prefix_ms", C.c_float),
                ("emit_ms", C.c_float), ("stats_ms", C.c_float)]


class IdxRecord(C.Structure):
    _fields_ = [("start", C.c_uint64), ("head", C.c_uint32), ("seq", C.c_uint32),
                ("sep", C.c_uint32), ("qual", C.c_uint32)]


class Chunk(C.Structure):
    _fields_ = [("parse_status", C.c_int32), ("is_final", C.c_int32), ("n_records", C.c_uint64),
                ("base_offset", C.c_uint64), ("data_len", C.c_uint64), ("lead_len", C.c_uint64),
                ("h_data", C.c_void_p), ("h_index", C.c_void_p), ("h_rec_start", C.c_void_p),
                ("d_data", C.c_void_p), ("d_rec_start", C.c_void_p),
                ("err_record", C.c_uint64), ("err_offset", C.c_uint64), ("err_need", C.c_uint64)]


class ShardResult(C.Structure):
    """fqh_shard_result: what one rank of the sharded, host-streamed mode found in its byte range."""
    _fields_ = [("status", C.c_int32), ("phase", C.c_uint32), ("n_records", C.c_uint64), ("n_newlines", C.c_uint64),
                ("err_offset", C.c_uint64), ("head_len", C.c_uint64), ("tail_len", C.c_uint64), ("flags", C.c_uint64)]


READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64)   # fqh_read_fn
MAP_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64))   # fqh_map_fn
SHARD_STREAM_WORDS = 10
SHARD_EMPTY, SHARD_PASS, SHARD_DEFER = 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFD
NO_ERROR_KEY = (1 << 64) - 1
STREAM_INDEX = 1
STREAM_STATS = 2
STREAM_TIMING = 4
STREAM_EXTERNAL = 8


class StreamTimes(C.Structure):
    _fields_ = [("wall_ms", C.c_double), ("copy_busy_ms", C.c_double), ("scan_busy_ms", C.c_double), ("both_busy_ms", C.c_double),
                ("n_slots", C.c_uint64)]


class FqhError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("fqh status %d: %s" % (status, msg))
        self.status = status


_LIB = None


def lib():
    """Loads libfastq_hip.so.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C fastq-rs_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # When torch is used in the same process (tests, bench.py) it must be imported BEFORE this
        # library is loaded: torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and the
        # dynamic linker then resolves our NEEDED libamdhip64.so.7 to that one instance.  Loaded the
        # other way round the process ends up with two HIP runtimes and fqh_create fails.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.fqh_create.argtypes = [i32, C.POINTER(vp)]
        L.fqh_destroy.argtypes = [vp]
        L.fqh_destroy.restype = None
        L.fqh_strerror.argtypes = [i32]
        L.fqh_strerror.restype = C.c_char_p
        L.fqh_last_error.argtypes = [vp]
        L.fqh_last_error.restype = C.c_char_p
        L.fqh_set_stream.argtypes = [vp, vp]
        L.fqh_set_bufsize.argtypes = [vp, u64]
        L.fqh_scan.argtypes = [vp, vp, u64, i32, C.POINTER(Carry), vp, u64, C.POINTER(Summary),
                               C.POINTER(Carry)]
        L.fqh_scan_launch.argtypes = [vp, vp, u64, i32, C.POINTER(Carry), vp, u64]
        L.fqh_scan_finish.argtypes = [vp, C.POINTER(Summary), C.POINTER(Carry)]
        L.fqh_index_records.argtypes = [vp, vp, u64]
        L.fqh_carry_combine.argtypes = [C.POINTER(Carry), u64, u64, u64, C.POINTER(u64 * 4),
                                        C.POINTER(Carry)]
        L.fqh_rescan_launch.argtypes = [vp, i32, C.POINTER(Carry), vp, u64]
        L.fqh_shard_prescan.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64 * 4)]
        L.fqh_shard_align.argtypes = [vp, vp, u64, i32, C.POINTER(u32), C.POINTER(u64)]
        L.fqh_shard_prescan_launch.argtypes = [vp, vp, u64, vp]
        L.fqh_shard_rescan_launch.argtypes = [vp, i32, vp, i32, i32, vp, u64, vp]
        L.fqh_stream_carry.argtypes = [vp, C.POINTER(Carry)]
        L.fqh_invalidate.argtypes = [vp]
        L.fqh_stats.argtypes = [vp, vp, u64, i32, C.POINTER(Carry), u32, vp, vp, vp,
                                C.POINTER(Summary), C.POINTER(Carry)]
        L.fqh_stats_launch.argtypes = [vp, vp, u64, i32, C.POINTER(Carry), u32, vp, vp, vp]
        L.fqh_stats_finish.argtypes = [vp, C.POINTER(Summary), C.POINTER(Carry)]
        L.fqh_stats_launch_lead.argtypes = [vp, vp, u64, u64, i32, C.POINTER(Carry), u32, vp, vp, vp]
        L.fqh_scan_stats.argtypes = [vp, vp, u64, i32, C.POINTER(Carry), vp, u64, u32, vp, vp, vp,
                                     C.POINTER(Summary), C.POINTER(Carry)]
        L.fqh_scan_stats_launch.argtypes = [vp, vp, u64, i32, C.POINTER(Carry), vp, u64, u32, vp, vp, vp]
        L.fqh_scan_stats_finish.argtypes = [vp, C.POINTER(Summary), C.POINTER(Carry)]
        L.fqh_stream_set_stats.argtypes = [vp, u32, vp, vp, vp]
        L.fqh_last_timing.argtypes = [vp, C.POINTER(Timing)]
        L.fqh_record_flags.argtypes = [vp, vp, u64, u64, vp, u64, vp]
        L.fqh_len_hist.argtypes = [vp, vp, vp, C.c_uint32, vp]
        L.fqh_gather_records.argtypes = [vp, vp, u64, u64, vp, u64, vp, C.c_uint8, C.c_uint8, vp, u64,
                                         C.POINTER(u64), C.POINTER(u64)]
        L.fqh_last_scan_fast.argtypes = [vp]
        L.fqh_last_stats_route.argtypes = [vp]
        L.fqh_set_option.argtypes = [vp, i32, i32]
        L.fqh_placement.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_float * 10)]
        L.fqh_line_buffers.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64)]
        L.fqh_stream_create.argtypes = [vp, u64, u32, u32, C.POINTER(vp)]
        L.fqh_stream_destroy.argtypes = [vp]
        L.fqh_stream_destroy.restype = None
        L.fqh_stream_acquire.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        L.fqh_stream_submit.argtypes = [vp, u64, i32]
        L.fqh_stream_submit_external.argtypes = [vp, vp, u64, i32]
        L.fqh_stream_release_chunk.argtypes = [vp, C.POINTER(Chunk)]
        L.fqh_host_register.argtypes = [vp, vp, u64]
        L.fqh_host_unregister.argtypes = [vp, vp]
        L.fqh_stream_collect.argtypes = [vp, C.POINTER(Chunk)]
        L.fqh_stream_release.argtypes = [vp]
        L.fqh_stream_timing.argtypes = [vp, C.POINTER(StreamTimes)]
        L.fqh_comm_unique_id.argtypes = [C.c_char_p]
        L.fqh_comm_create.argtypes = [vp, i32, i32, C.c_char_p, C.POINTER(vp)]
        L.fqh_comm_destroy.argtypes = [vp]
        L.fqh_comm_destroy.restype = None
        L.fqh_allgather.argtypes = [vp, vp, vp, vp, u64]
        L.fqh_allreduce_u64.argtypes = [vp, vp, vp, u64]
        L.fqh_allreduce_min_u64.argtypes = [vp, vp, vp, u64]
        L.fqh_shard_stream_run.argtypes = [vp, READ_FN, vp, u64, u64, u64, u64, u32, u32, vp, vp, vp, C.POINTER(ShardResult)]
        L.fqh_shard_stream_run_mapped.argtypes = [vp, READ_FN, MAP_FN, vp, u64, u64, u64, u64, u32, u32, vp, vp, vp, C.POINTER(ShardResult)]
        L.fqh_shard_result_words.argtypes = [C.POINTER(ShardResult), u64, u64, C.POINTER(u64 * SHARD_STREAM_WORDS)]
        L.fqh_shard_result_words.restype = None
        L.fqh_shard_failed_words.argtypes = [i32, u64, u64, C.POINTER(u64 * SHARD_STREAM_WORDS)]
        L.fqh_shard_failed_words.restype = None
        L.fqh_shard_failure_key.argtypes = [i32, u64, i32]
        L.fqh_shard_failure_key.restype = u64
        L.fqh_shard_stream_finish.argtypes = [vp, READ_FN, vp, u64, vp, i32, i32, u64, u32, u32, vp, vp, vp, C.POINTER(u64 * 2)]
        L.fqh_shard_stream_outcome.argtypes = [u64, vp, i32, C.POINTER(C.c_int32), C.POINTER(u64), C.POINTER(u64)]
        L.fqh_stream_set_origin.argtypes = [vp, u64]
        L.fqh_stream_note_read.argtypes = [vp, u64, u64]
        L.fqh_sync.argtypes = [vp]
        L.fqh_synth_fill.argtypes = [vp, vp, u64, u64, u64]
        L.fqh_read_ceiling.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(C.c_float)]
        L.fqh_dev_alloc.argtypes = [vp, u64, C.POINTER(vp)]
        L.fqh_dev_free.argtypes = [vp, vp]
        L.fqh_memcpy_h2d.argtypes = [vp, vp, vp, u64]
        L.fqh_memcpy_d2h.argtypes = [vp, vp, vp, u64]
        L.fqh_memset.argtypes = [vp, vp, i32, u64]
        _LIB = L
    return _LIB


def strerror(status):
    return lib().fqh_strerror(status).decode()


def shard_stream_outcome(min_key, records_per_rank):
    """The two reductions of the sharded, host-streamed mode -> Parser::each's result (fqh_shard_stream_outcome):
    (status, records delivered before the first error — all of them when status is OK —, file offset of the failing record)."""
    n = len(records_per_rank)
    arr = (C.c_uint64 * n)(*[int(x) for x in records_per_rank])
    st, rec, off = C.c_int32(), C.c_uint64(), C.c_uint64()
    rc = lib().fqh_shard_stream_outcome(min_key, C.addressof(arr), n, C.byref(st), C.byref(rec), C.byref(off))
    if rc != OK:
        raise FqhError(rc, "fqh_shard_stream_outcome")
    return st.value, rec.value, off.value


def shard_failed_words(status, lo, hi):
    """The words a rank sends into the exchange when its fqh_shard_stream_run failed (it must still take part)."""
    w = (C.c_uint64 * SHARD_STREAM_WORDS)()
    lib().fqh_shard_failed_words(status, lo, hi, C.byref(w))
    return [int(x) for x in w]


def shard_failure_key(rank, offset, status):
    return int(lib().fqh_shard_failure_key(rank, offset, status))


def carry_combine(prev, length, n_newlines, n_line_starts, back_zero_carry):
    """Folds one shard's zero-carry summary into the running carry (host-only, no GPU)."""
    nxt = Carry()
    arr = (C.c_uint64 * 4)(*[int(x) for x in back_zero_carry])
    st = lib().fqh_carry_combine(C.byref(prev) if prev is not None else None, length, n_newlines,
                                 n_line_starts, C.byref(arr), C.byref(nxt))
    if st != OK:
        raise FqhError(st, "fqh_carry_combine")
    return nxt


class Ctx:
    """One fqh_ctx.  Pointers are raw device addresses (ints)."""

    def __init__(self, device=0, stream=None, bufsize=None):
        self._L = lib()
        h = C.c_void_p()
        st = self._L.fqh_create(device, C.byref(h))
        if st != OK:
            raise FqhError(st, "fqh_create failed: " + self._L.fqh_last_error(None).decode())
        self._h = h
        if stream is not None:
            self._chk(self._L.fqh_set_stream(self._h, C.c_void_p(stream)))
        if bufsize is not None:
            self.set_bufsize(bufsize)

    def close(self):
        if self._h:
            self._L.fqh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, allow=()):
        if st != OK and st not in allow:
            raise FqhError(st, self._L.fqh_last_error(self._h).decode() or strerror(st))
        return st

    def set_stream(self, stream):
        self._chk(self._L.fqh_set_stream(self._h, C.c_void_p(stream or 0)))

    def set_bufsize(self, bufsize):
        self._chk(self._L.fqh_set_bufsize(self._h, bufsize))

    def scan(self, d_buf, length, is_final=True, carry=None, d_rec_start=None, cap=0):
        s, c = Summary(), Carry()
        st = self._L.fqh_scan(self._h, d_buf, length, 1 if is_final else 0,
                              C.byref(carry) if carry is not None else None,
                              d_rec_start, cap, C.byref(s), C.byref(c))
        self._chk(st, allow=(E_CAPACITY,))
        return s, c, st

    def scan_launch(self, d_buf, length, is_final=True, carry=None, d_rec_start=None, cap=0):
        self._chk(self._L.fqh_scan_launch(self._h, d_buf, length, 1 if is_final else 0,
                                          C.byref(carry) if carry is not None else None,
                                          d_rec_start, cap))

    def scan_finish(self):
        s, c = Summary(), Carry()
        st = self._L.fqh_scan_finish(self._h, C.byref(s), C.byref(c))
        self._chk(st, allow=(E_CAPACITY,))
        return s, c, st

    def shard_prescan(self, d_buf, length):
        """-> (n_newlines, n_line_starts, back[4]) of the shard scanned as if it began the file."""
        nn, ns, back = C.c_uint64(), C.c_uint64(), (C.c_uint64 * 4)()
        self._chk(self._L.fqh_shard_prescan(self._h, d_buf, length, C.byref(nn), C.byref(ns), C.byref(back)))
        return nn.value, ns.value, [int(x) for x in back]

    def shard_prescan_launch(self, d_buf, length, d_words):
        """The byte scan of a shard, enqueued; its SHARD_WORDS words go to d_words (device) for the all-gather."""
        self._chk(self._L.fqh_shard_prescan_launch(self._h, d_buf, length, d_words))

    def shard_rescan_launch(self, is_final, d_all_words, n_ranks, rank, d_rec_start=None, cap=0, d_counts=None):
        """Fold of the gathered words + emit under the true carry, enqueued; end with scan_finish() (raises FqhError with
        status E_AGAIN on every rank if a shard left the fast path: take the host recipe then)."""
        self._chk(self._L.fqh_shard_rescan_launch(self._h, 1 if is_final else 0, d_all_words, n_ranks, rank, d_rec_start, cap,
                                                  d_counts))

    def shard_align(self, d_buf, length, prev_is_newline):
        """-> (phase, first_record_offset) of a shard that starts anywhere; raises FqhError(E_HEADER / E_ARG) if the window
        does not settle it."""
        ph, off = C.c_uint32(), C.c_uint64()
        self._chk(self._L.fqh_shard_align(self._h, d_buf, length, 1 if prev_is_newline else 0, C.byref(ph), C.byref(off)))
        return ph.value, off.value

    def rescan_launch(self, is_final=True, carry=None, d_rec_start=None, cap=0):
        self._chk(self._L.fqh_rescan_launch(self._h, 1 if is_final else 0,
                                            C.byref(carry) if carry is not None else None,
                                            d_rec_start, cap))

    def invalidate(self):
        self._chk(self._L.fqh_invalidate(self._h))

    def index_records(self, d_index, cap):
        self._chk(self._L.fqh_index_records(self._h, d_index, cap))

    def stats(self, d_buf, length, lmax, d_qual, d_base, d_scalars, is_final=True, carry=None):
        s, c = Summary(), Carry()
        self._chk(self._L.fqh_stats(self._h, d_buf, length, 1 if is_final else 0,
                                    C.byref(carry) if carry is not None else None, lmax,
                                    d_qual, d_base, d_scalars, C.byref(s), C.byref(c)))
        return s, c

    def len_hist(self, d_base_hist, d_scalars, lmax, d_len_hist):
        """d_len_hist[L] += reads with len(seq()) == L (L < lmax), d_len_hist[lmax] += reads of lmax bases or more."""
        self._chk(self._L.fqh_len_hist(self._h, d_base_hist, d_scalars, lmax, d_len_hist))

    def record_flags(self, d_buf, length, d_index, n, d_flags, base_offset=0):
        self._chk(self._L.fqh_record_flags(self._h, d_buf, length, base_offset, d_index, n, d_flags))

    def gather_records(self, d_buf, length, d_index, n, d_flags, mask, want, d_out, out_cap, base_offset=0):
        """-> (status, n_selected, out_bytes); status is OK or E_CAPACITY."""
        ns, nb = C.c_uint64(), C.c_uint64()
        st = self._L.fqh_gather_records(self._h, d_buf, length, base_offset, d_index, n, d_flags, mask, want,
                                        d_out, out_cap, C.byref(ns), C.byref(nb))
        if st not in (OK, E_CAPACITY):
            self._chk(st)
        return st, ns.value, nb.value

    def stats_launch(self, d_buf, length, lmax, d_qual, d_base, d_scalars, is_final=True, carry=None):
        self._chk(self._L.fqh_stats_launch(self._h, d_buf, length, 1 if is_final else 0,
                                           C.byref(carry) if carry is not None else None, lmax,
                                           d_qual, d_base, d_scalars))

    def stats_launch_lead(self, d_buf, length, lead_len, lmax, d_qual, d_base, d_scalars, is_final=True, carry=None):
        """d_buf[-lead_len:] is valid device memory holding the start of the record in progress."""
        self._chk(self._L.fqh_stats_launch_lead(self._h, d_buf, length, lead_len, 1 if is_final else 0,
                                                C.byref(carry) if carry is not None else None, lmax,
                                                d_qual, d_base, d_scalars))

    def scan_stats(self, d_buf, length, lmax, d_qual, d_base, d_scalars, is_final=True, carry=None,
                   d_rec_start=None, cap=0):
        """Offsets + histograms in one call (one read of the input for a whole file) -> (summary, carry, status)."""
        s, c = Summary(), Carry()
        st = self._L.fqh_scan_stats(self._h, d_buf, length, 1 if is_final else 0,
                                    C.byref(carry) if carry is not None else None, d_rec_start, cap, lmax,
                                    d_qual, d_base, d_scalars, C.byref(s), C.byref(c))
        self._chk(st, allow=(E_CAPACITY,))
        return s, c, st

    def scan_stats_launch(self, d_buf, length, lmax, d_qual, d_base, d_scalars, is_final=True, carry=None,
                          d_rec_start=None, cap=0):
        self._chk(self._L.fqh_scan_stats_launch(self._h, d_buf, length, 1 if is_final else 0,
                                                C.byref(carry) if carry is not None else None, d_rec_start, cap,
                                                lmax, d_qual, d_base, d_scalars))

    def stats_finish(self):
        s, c = Summary(), Carry()
        self._chk(self._L.fqh_stats_finish(self._h, C.byref(s), C.byref(c)))
        return s, c

    def last_scan_fast(self):
        """Test hook: did the last scan complete on the fast path (no exact rerun)?"""
        return bool(self._L.fqh_last_scan_fast(self._h))

    def last_stats_route(self):
        """How the last statistics call counted: 1 single pass, 2 single pass + declined lines recounted, 0 second pass."""
        return int(self._L.fqh_last_stats_route(self._h))

    def set_spec(self, on):
        """Test hook: (re-)enable or disable the fast path for this context."""
        self._chk(self._L.fqh_set_option(self._h, OPT_FAST_PATH, 1 if on else 0))

    def set_single_pass(self, on):
        """Whole-file statistics in the scan's own pass over the input (default) or as a second pass."""
        self._chk(self._L.fqh_set_option(self._h, OPT_SINGLE_PASS, 1 if on else 0))

    def set_place_tries(self, n):
        """Candidates of the fast path's per-tile line buffer the first big scan allocates and times (0 / 1: none)."""
        self._chk(self._L.fqh_set_option(self._h, OPT_PLACE_TRIES, int(n)))

    def set_adapt_lines(self, n):
        """Alternate line buffers a context may try per big input it is given again (FQH_OPT_ADAPT_LINES; 0: off)."""
        self._chk(self._L.fqh_set_option(self._h, OPT_ADAPT_LINES, int(n)))

    def set_own_stream_nonblocking(self, on):
        """The context's own stream as a non-blocking one (no ordering against the legacy null stream); default: blocking."""
        self._chk(self._L.fqh_set_option(self._h, OPT_OWN_STREAM_NONBLOCKING, 1 if on else 0))

    def set_reuse_index(self, on):
        """Let fqh_stats* count over the last scan's tile index when buffer, length and carry match (the caller vouches for the bytes)."""
        self._chk(self._L.fqh_set_option(self._h, OPT_REUSE_INDEX, 1 if on else 0))

    def set_keep_ring(self, on):
        """A destroyed ring's pinned slots stay with the context for the next ring of the same geometry (FQH_OPT_KEEP_RING)."""
        self._chk(self._L.fqh_set_option(self._h, OPT_KEEP_RING, 1 if on else 0))

    def host_register(self, addr, nbytes):
        """Page-locks the caller's own host memory (a source of Stream.submit_external / a mapped shard run)."""
        self._chk(self._L.fqh_host_register(self._h, addr, nbytes))

    def host_unregister(self, addr):
        self._chk(self._L.fqh_host_unregister(self._h, addr))

    def set_spin_wait(self, usec):
        """Microseconds *_finish polls the stream before sleeping on it (default 0: sleeps at once)."""
        self._chk(self._L.fqh_set_option(self._h, OPT_SPIN_WAIT, int(usec)))

    def placement(self):
        """What the placement search of this context measured -> dict (engaged False: no search ran)."""
        n, ms = C.c_int(0), (C.c_float * 10)()
        self._chk(self._L.fqh_placement(self._h, C.byref(n), C.byref(ms)))
        return {"engaged": n.value > 0, "candidates": n.value, "candidate_ms": [round(float(ms[i]), 4) for i in range(n.value)],
                "no_store_ms": round(float(ms[9]), 4), "best_ms": round(float(ms[8]), 4)}

    def line_buffers(self):
        """What FQH_OPT_ADAPT_LINES holds -> dict(alive, unsettled, bytes)."""
        n, u, b = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
        self._chk(self._L.fqh_line_buffers(self._h, C.byref(n), C.byref(u), C.byref(b)))
        return {"alive": n.value, "unsettled": u.value, "bytes": b.value}

    def timing(self):
        t = Timing()
        self._chk(self._L.fqh_last_timing(self._h, C.byref(t)))
        return t

    def synth_fill(self, d_out, byte_off, length, seed=0x5EEDF00D2026):
        self._chk(self._L.fqh_synth_fill(self._h, d_out, byte_off, length, seed))

    def read_ceiling(self, d_buf, length):
        cs, ms = C.c_uint64(0), C.c_float(0)
        self._chk(self._L.fqh_read_ceiling(self._h, d_buf, length, C.byref(cs), C.byref(ms)))
        return cs.value, ms.value


COMM_ID_BYTES = 128


class Comm:
    """fqh_comm_*: the library's own RCCL binding (csrc/comm.hip) — the exchange steps of the sharded modes for hosts without a
    collective library; every call is enqueued on the context's stream.  The gather of Parser::parallel_each's results,
    src/lib.rs:553-559, and the parse error it returns, src/lib.rs:544-547, 561-564."""

    @staticmethod
    def unique_id():
        """One rank makes the id (-> 128 bytes) and hands it to the others by any means."""
        uid = C.create_string_buffer(COMM_ID_BYTES)
        st = lib().fqh_comm_unique_id(uid)
        if st != OK:
            raise FqhError(st, "fqh_comm_unique_id: librccl.so could not be loaded or ncclGetUniqueId failed")
        return uid.raw

    def __init__(self, ctx, n_ranks, rank, uid):
        self.ctx = ctx
        self._L = ctx._L
        self.n_ranks, self.rank = n_ranks, rank
        h = C.c_void_p()
        ctx._chk(self._L.fqh_comm_create(ctx._h, n_ranks, rank, C.create_string_buffer(bytes(uid), COMM_ID_BYTES), C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._L.fqh_comm_destroy(self._h)
            self._h = None

    def allgather(self, d_send, d_recv, bytes_per_rank):
        self.ctx._chk(self._L.fqh_allgather(self.ctx._h, self._h, d_send, d_recv, bytes_per_rank))

    def allreduce_u64(self, d_buf, n):
        self.ctx._chk(self._L.fqh_allreduce_u64(self.ctx._h, self._h, d_buf, n))

    def allreduce_min_u64(self, d_buf, n):
        self.ctx._chk(self._L.fqh_allreduce_min_u64(self.ctx._h, self._h, d_buf, n))

    def sync(self):
        self.ctx._chk(self._L.fqh_sync(self.ctx._h))


class Stream:
    """fqh_stream_*: pinned ring + overlapped H2D in front of the scan (single producer/consumer)."""

    def __init__(self, ctx, slot_bytes, n_slots=3, flags=STREAM_INDEX):
        self.ctx = ctx
        self._L = ctx._L
        h = C.c_void_p()
        ctx._chk(self._L.fqh_stream_create(ctx._h, slot_bytes, n_slots, flags, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._L.fqh_stream_destroy(self._h)
            self._h = None

    def set_stats(self, lmax, d_qual, d_base, d_scalars):
        self.ctx._chk(self._L.fqh_stream_set_stats(self._h, lmax, d_qual, d_base, d_scalars))

    def acquire(self):
        """-> (host address, capacity) or None when the ring is full."""
        p, cap = C.c_void_p(), C.c_uint64()
        st = self._L.fqh_stream_acquire(self._h, C.byref(p), C.byref(cap))
        if st == E_CAPACITY:
            return None
        self.ctx._chk(st)
        return p.value, cap.value

    def acquire_status(self):
        """The raw status of fqh_stream_acquire (for tests of its error paths); a slot it hands out is given back empty."""
        p, cap = C.c_void_p(), C.c_uint64()
        st = self._L.fqh_stream_acquire(self._h, C.byref(p), C.byref(cap))
        if st == OK:
            self.ctx._chk(self._L.fqh_stream_submit(self._h, 0, 0))
        return st

    def note_read(self, got, asked):
        """One read() of the host's reader into the acquired slot: got of asked bytes (a reader that may come back short)."""
        self.ctx._chk(self._L.fqh_stream_note_read(self._h, got, asked))

    def submit(self, nbytes, is_final):
        self.ctx._chk(self._L.fqh_stream_submit(self._h, nbytes, 1 if is_final else 0))

    def submit_external(self, host_addr, nbytes, is_final):
        """The next slot's bytes straight from the caller's (page-locked) memory -> True, or False when the ring is full."""
        st = self._L.fqh_stream_submit_external(self._h, host_addr, nbytes, 1 if is_final else 0)
        if st == E_CAPACITY:
            return False
        self.ctx._chk(st)
        return True

    def collect(self):
        """-> the next chunk, or None while the slot behind it is still held (FQH_E_AGAIN: release that chunk, call again)."""
        c = Chunk()
        st = self._L.fqh_stream_collect(self._h, C.byref(c))
        if st == E_AGAIN:
            return None
        self.ctx._chk(st)
        return c

    def release(self):
        self.ctx._chk(self._L.fqh_stream_release(self._h))

    def release_chunk(self, chunk):
        """Done with THIS chunk (several may be held at once, given back in any order)."""
        self.ctx._chk(self._L.fqh_stream_release_chunk(self._h, C.byref(chunk)))

    def carry(self):
        c = Carry()
        self.ctx._chk(self._L.fqh_stream_carry(self._h, C.byref(c)))
        return c

    def timing(self):
        """STREAM_TIMING: copy / scan busy times and their overlap over the slots collected so far."""
        t = StreamTimes()
        self.ctx._chk(self._L.fqh_stream_timing(self._h, C.byref(t)))
        return t
