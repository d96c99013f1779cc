"""parse_path's reader chain in the C++ mirror (fastq-rs_amd/host/fastq.hpp: with_plain_reader), the
counterpart of src/lib.rs:167-196: five-byte sniff, magic numbers, multi-member gzip decoded on a
thread_reader thread.  No GPU is touched: `host_tests --plain` stops before the Parser.  The GPU
half (fastq_count on .gz input == oracle count) is in tests/test_gpu_host_mirror.py."""
import bz2
import gzip
import os
import subprocess

import numpy as np
import pytest

import fuzzgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fastq-rs_amd", "host", "bin", "host_tests")


@pytest.fixture(scope="module")
def host_tests():
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build()
    return BIN


def fnv1a(b):
    h = 1469598103934665603
    for x in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def plain(host_tests, path):
    out = subprocess.run([host_tests, "--plain", str(path)], capture_output=True, text=True, timeout=120)
    return out.returncode, out.stdout.strip()


def test_plain_and_gzip_give_the_same_bytes(host_tests, tmp_path):
    rng = np.random.default_rng(77)
    data = fuzzgen.valid_file(rng, 3000, maxlen=150)
    want = "plain %d %016x" % (len(data), fnv1a(data))
    (tmp_path / "a.fq").write_bytes(data)
    (tmp_path / "a.fq.gz").write_bytes(gzip.compress(data))
    cut = len(data) // 3
    (tmp_path / "multi.fq.gz").write_bytes(gzip.compress(data[:cut]) + gzip.compress(data[cut:], 1))
    for name in ("a.fq", "a.fq.gz", "multi.fq.gz"):
        assert plain(host_tests, tmp_path / name) == (0, want), name


def test_sniffing_errors(host_tests, tmp_path):
    data = b"@a\nACGT\n+\nIIII\n" * 100
    (tmp_path / "short").write_bytes(b"@a\nA")                      # niffler: FileTooShort
    rc, out = plain(host_tests, tmp_path / "short")
    assert rc == 3 and "less than five bytes" in out
    (tmp_path / "x.bz2").write_bytes(bz2.compress(data))            # detected, no decoder in this build
    rc, out = plain(host_tests, tmp_path / "x.bz2")
    assert rc == 3 and "bzip2" in out
    (tmp_path / "x.xz").write_bytes(bytes([0xfd, 0x37, 0x7a, 0x58, 0x5a, 0]) + b"junk")
    rc, out = plain(host_tests, tmp_path / "x.xz")
    assert rc == 3 and "xz" in out
    (tmp_path / "x.zst").write_bytes(bytes([0x28, 0xb5, 0x2f, 0xfd, 0]) + b"junk")
    rc, out = plain(host_tests, tmp_path / "x.zst")
    assert rc == 3 and "zstd" in out
    z = gzip.compress(fuzzgen.valid_file(np.random.default_rng(5), 2000, maxlen=150))
    (tmp_path / "trunc.gz").write_bytes(z[:len(z) // 2])              # error surfaces, no hang
    rc, out = plain(host_tests, tmp_path / "trunc.gz")
    assert rc == 3 and out.startswith("error")
    (tmp_path / "garbage.gz").write_bytes(gzip.compress(data) + b"not a gzip member")
    rc, out = plain(host_tests, tmp_path / "garbage.gz")
    assert rc == 3 and out.startswith("error")
