"""Random FASTQ-ish inputs for differential tests (valid files, mutated files, raw garbage).

Mirrors the idea of the reference's fuzz targets (fuzz/fuzz_targets/fuzz_target_1.rs:11-18: arbitrary
bytes through Parser::each with BUFSIZE shrunk to 64, src/lib.rs:126-127), but differential: every
consumer compares two implementations on the same bytes.
"""
import numpy as np

ALPH = np.frombuffer(b"ACGTN", dtype=np.uint8)


def valid_record(rng, i, maxlen=40, crlf=False, plus_id=False, seqlen=None):
    n = int(rng.integers(0, maxlen + 1)) if seqlen is None else seqlen
    hid = b"r%d" % i + bytes(rng.integers(33, 127, int(rng.integers(0, 6))).astype(np.uint8).tolist())
    seq = bytes(rng.choice(ALPH, n).tolist()) if rng.random() < 0.9 else \
        bytes(rng.integers(33, 127, n).astype(np.uint8).tolist())
    qual = bytes(rng.integers(33, 75, n).astype(np.uint8).tolist())
    e = b"\r\n" if crlf else b"\n"
    sep = b"+" + (hid if plus_id else b"")
    return b"@" + hid + e + seq + e + sep + e + qual + e


def valid_file(rng, nrec, **kw):
    out = []
    for i in range(nrec):
        out.append(valid_record(rng, i, crlf=kw.get("crlf", rng.random() < 0.1),
                                plus_id=rng.random() < 0.2, maxlen=kw.get("maxlen", 40),
                                seqlen=kw.get("seqlen")))
    return b"".join(out)


def mutate(rng, data, nmut=1):
    b = bytearray(data)
    for _ in range(nmut):
        if not b:
            break
        op = int(rng.integers(0, 6))
        pos = int(rng.integers(0, len(b)))
        if op == 0:
            del b[pos]
        elif op == 1:
            b.insert(pos, int(rng.choice(list(b"\n\r@+ACGT!I~"))))
        elif op == 2:
            b[pos] = int(rng.choice(list(b"\n\r@+ACGT!I~")))
        elif op == 3:
            del b[pos:]
        elif op == 4 and b[pos] == 10:
            b[pos:pos + 1] = b"\r\n"
        else:
            b[pos] = int(rng.integers(0, 256))
    return bytes(b)


def garbage(rng, n):
    # newline / sentinel heavy garbage reaches deep parser states quickly
    pool = list(b"\n\n\n@@++\rACGTI!") + [int(rng.integers(0, 256))]
    return bytes(rng.choice(pool, n).astype(np.uint8).tolist())


def corpus(seed, n):
    """Yields (tag, bytes)."""
    rng = np.random.default_rng(seed)
    for i in range(n):
        r = rng.random()
        nrec = int(rng.integers(0, 30))
        if r < 0.25:
            yield "valid", valid_file(rng, nrec)
        elif r < 0.75:
            yield "mutated", mutate(rng, valid_file(rng, nrec), int(rng.integers(1, 4)))
        elif r < 0.85:
            yield "garbage", garbage(rng, int(rng.integers(0, 200)))
        elif r < 0.95:
            # a few long records between short ones: exercises the too-long band at BUFSIZE=64
            parts = []
            for j in range(nrec):
                L = int(rng.integers(0, 40)) if rng.random() < 0.7 else int(rng.integers(14, 30))
                parts.append(valid_record(rng, j, seqlen=L))
            yield "longish", b"".join(parts)
        else:
            yield "mutated-long", mutate(rng, valid_file(rng, nrec, maxlen=80), 1)
