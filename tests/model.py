"""Whole-buffer model of the scan path — the SPEC the HIP kernels implement.

The reference parses one record at a time through a 68 KiB window.  Because it accepts only strict
4-line records (src/records.rs:201-247), the same result is a function of the positions of ALL
newlines: record k is lines 4k..4k+3, and validity is three local predicates + an EOF rule
(SURVEY §7.0).  This file states that function in numpy/Python; tests/test_model_vs_oracle.py
proves it equal to the streaming oracle on thousands of random inputs (also with BUFSIZE=64, the
reference's cfg(fuzzing) value), and the GPU parity tests compare the kernels with the oracle.

It is test code (a second, independent statement of the semantics), not product code.
"""
import bisect

import numpy as np

OK, E_HEADER, E_SEP, E_LEN, E_TRUNCATED, E_TOO_LONG = range(6)
STAGE_TO_STATUS = {0: E_HEADER, 1: E_SEP, 2: E_LEN, 3: E_TRUNCATED}
NOKEY = 1 << 62


def newline_index(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    return a, np.flatnonzero(a == 10).astype(np.int64)


def first_error_key(a, nl, is_final=True):
    """min over all violated predicates of key = record*4 + stage
    (stage 0 header-'@', 1 sep-'+', 2 length mismatch, 3 truncated)."""
    N, T = a.size, nl.size
    key = NOKEY
    # line starts: byte 0 (line 0) and the byte after newline j (line j+1), when that byte exists
    starts = np.concatenate(([0], nl + 1)) if N else np.zeros(0, np.int64)
    lines = np.arange(starts.size)
    ok = starts < N
    starts, lines = starts[ok], lines[ok]
    first = a[starts] if starts.size else np.zeros(0, np.uint8)
    bad = lines[(lines % 4 == 0) & (first != ord("@"))]
    if bad.size:
        key = min(key, int(bad[0] // 4) * 4 + 0)
    bad = lines[(lines % 4 == 2) & (first != ord("+"))]
    if bad.size:
        key = min(key, int(bad[0] // 4) * 4 + 1)
    K = T // 4
    if K:
        q = nl[: 4 * K].reshape(K, 4)
        mism = np.flatnonzero((q[:, 1] - q[:, 0]) != (q[:, 3] - q[:, 2]))
        if mism.size:
            key = min(key, int(mism[0]) * 4 + 2)
    end_complete = int(nl[4 * K - 1]) + 1 if K else 0
    if is_final and end_complete < N:
        key = min(key, K * 4 + 3)
    return key


def rec_starts(nl, K):
    """rec_start[0..K]: start of every complete record and the end of the last one."""
    rs = np.zeros(K + 1, dtype=np.int64)
    if K:
        rs[1:] = nl[3: 4 * K: 4] + 1
    return rs


def visible_need(a, nl, e, stage):
    """Bytes of record e (from its start) the parser must see before it can report `stage`."""
    N = a.size
    rs = int(nl[4 * e - 1]) + 1 if e else 0
    if stage == 0:
        return 1
    if stage == 1:
        return int(nl[4 * e + 1]) + 2 - rs
    if stage == 2:
        return int(nl[4 * e + 3]) + 1 - rs
    return None  # truncated: only reported once read() returns 0


def resolve_too_long(rs, n_valid, N, err_need, bufsize):
    """Exact replay of the Buffer arithmetic (src/buffer.rs:51-100) at refill granularity for
    Parser::each (src/lib.rs:255-303), driven only by record boundaries.

    rs[0..n_valid] record boundaries of the valid records (rs[n_valid] = end of the last one);
    err_need = bytes of the first non-valid record needed to report its error (None: truncated tail,
    -1: there is no such record, the input ends at rs[n_valid]).
    Returns (too_long, record_index).
    """
    B = bufsize
    start = end = 0
    fpos = 0       # file offset of buffer[start]
    rd = 0         # file offset of buffer[end] (bytes read so far)
    k = 0          # index of the record beginning at fpos
    while True:
        # consume every complete valid record that lies inside the window [fpos, rd)
        j = bisect.bisect_right(rs, rd, lo=k, hi=n_valid + 1) - 1
        if j > k:
            start += int(rs[j]) - fpos
            fpos = int(rs[j])
            k = j
        if k == n_valid:
            if err_need == -1 and fpos == N and start == end:
                return False, k   # EmptyBuffer at EOF
            if err_need is not None and err_need != -1 and fpos + err_need <= rd:
                return False, k   # the error is visible in the window: reported as itself
        if start == end:          # EmptyBuffer: clean(); read_into()
            if start:
                start = end = 0
        else:                     # Incomplete: clean(); n_free()==0 => too long
            if start:
                n = end - start
                new_end = (n + 15) & ~15
                new_start = new_end - n
                if new_start < start:
                    start, end = new_start, new_end
            if B - end == 0:
                return True, k
        n_free = B - end
        num = n_free if n_free < 4096 else n_free - n_free % 4096
        got = min(num, N - rd)
        if got == 0:
            return False, k       # EOF: Ok(end) or "truncated", decided by the caller's key
        end += got
        rd += got


def scan(data, is_final=True, bufsize=None):
    """-> dict(status, n_records, rec_start (np.int64[n_records+1]), err_record)."""
    a, nl = newline_index(data)
    N = a.size
    key = first_error_key(a, nl, is_final)
    K = nl.size // 4
    if key == NOKEY:
        n_valid, status, stage = K, OK, None
    else:
        n_valid, stage = key // 4, key % 4
        status = STAGE_TO_STATUS[stage]
    rs = rec_starts(nl, K)[: n_valid + 1]
    if bufsize is not None and is_final:
        if key == NOKEY:
            need = -1
        else:
            need = visible_need(a, nl, n_valid, stage)
        too_long, k = resolve_too_long(rs, n_valid, N, need, bufsize)
        if too_long:
            status, n_valid = E_TOO_LONG, k
            rs = rs[: k + 1]
    return {"status": status, "n_records": n_valid, "rec_start": rs,
            "err_record": n_valid if status != OK else None}


def stats(data, lmax, is_final=True, bufsize=None):
    """Histograms over the records scan() delivers.  -> (scan dict, qual_hist, base_hist, scalars)."""
    s = scan(data, is_final, bufsize)
    a, nl = newline_index(data)
    qh = np.zeros((lmax, 256), dtype=np.uint64)
    bh = np.zeros((lmax, 8), dtype=np.uint64)
    sc = np.zeros(8, dtype=np.uint64)
    cls = np.full(256, 5, dtype=np.int64)
    for i, c in enumerate(b"ACGTN"):
        cls[c] = i
    for k in range(s["n_records"]):
        n0, n1, n2, n3 = (int(x) for x in nl[4 * k: 4 * k + 4])
        seq = a[n0 + 1: n1]
        qual = a[n2 + 1: n3]
        if seq.size and seq[-1] == 13:
            seq = seq[:-1]
        if qual.size and qual[-1] == 13:
            qual = qual[:-1]
        m = min(lmax, seq.size)
        np.add.at(bh, (np.arange(m), cls[seq[:m]]), 1)
        m = min(lmax, qual.size)
        np.add.at(qh, (np.arange(m), qual[:m]), 1)
        sc[0] += 1
        sc[1] += seq.size
        sc[2] += qual.size
        c = cls[seq]
        sc[3] += int(np.all(c < 4))
        sc[4] += int(np.all(c < 5))
        sc[5] += max(0, seq.size - lmax)
        sc[6] += max(0, qual.size - lmax)
    return s, qh, bh, sc
