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


def lz4_frame(data, level=0, block_linked=False, content_checksum=False, block_size_id=0):
    """An LZ4 frame made by the system's liblz4 (LZ4F_compressFrame) — an encoder that is not ours."""
    import ctypes as C
    L = C.CDLL("liblz4.so.1")

    class FrameInfo(C.Structure):
        _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int), ("frameType", C.c_int),
                    ("contentSize", C.c_ulonglong), ("dictID", C.c_uint), ("blockChecksumFlag", C.c_int)]

    class Prefs(C.Structure):
        _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                    ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]

    pr = Prefs()
    pr.frameInfo.blockSizeID = block_size_id
    pr.frameInfo.blockMode = 0 if block_linked else 1   # LZ4F_blockLinked = 0, LZ4F_blockIndependent = 1
    pr.frameInfo.contentChecksumFlag = 1 if content_checksum else 0
    pr.compressionLevel = level
    L.LZ4F_compressFrameBound.restype = C.c_size_t
    L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(Prefs)]
    L.LZ4F_compressFrame.restype = C.c_size_t
    L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Prefs)]
    L.LZ4F_isError.argtypes = [C.c_size_t]
    cap = L.LZ4F_compressFrameBound(len(data), C.byref(pr))
    out = C.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(out, cap, data, len(data), C.byref(pr))
    assert not L.LZ4F_isError(n)
    return out.raw[:n]


def test_lz4_frames_give_the_same_bytes(host_tests, tmp_path):
    """The crate's documentation names lz4 next to gzip (src/lib.rs:137-141, README.md:39-45).  Frames written by the
    system's liblz4: linked and independent blocks, 64 KiB and 4 MiB blocks, a content checksum, two frames back to
    back with a skippable frame between them; the decoder of parse_path gives back the plain bytes."""
    try:
        lz4_frame(b"x" * 100)
    except OSError:
        pytest.skip("no liblz4.so.1 on this box to write test frames with")
    rng = np.random.default_rng(78)
    data = fuzzgen.valid_file(rng, 6000, maxlen=150)
    want = "plain %d %016x" % (len(data), fnv1a(data))
    cut = len(data) // 3
    skippable = bytes([0x53, 0x2a, 0x4d, 0x18, 5, 0, 0, 0]) + b"hello"
    variants = {
        "a.lz4": lz4_frame(data),
        "linked.lz4": lz4_frame(data, block_linked=True, block_size_id=4),
        "hc.lz4": lz4_frame(data, level=9, block_linked=True, content_checksum=True, block_size_id=7),
        "multi.lz4": lz4_frame(data[:cut]) + skippable + lz4_frame(data[cut:], block_linked=True, block_size_id=4),
    }
    for name, blob in variants.items():
        (tmp_path / name).write_bytes(blob)
        assert plain(host_tests, tmp_path / name) == (0, want), name
    (tmp_path / "trunc.lz4").write_bytes(variants["a.lz4"][: len(variants["a.lz4"]) // 2])
    rc, out = plain(host_tests, tmp_path / "trunc.lz4")
    assert rc == 3 and out.startswith("error")


def zstd_frame(data, level=3):
    """A zstd frame made by the system's libzstd (ZSTD_compress)."""
    import ctypes as C
    L = C.CDLL("libzstd.so.1")
    L.ZSTD_compressBound.restype = C.c_size_t
    L.ZSTD_compressBound.argtypes = [C.c_size_t]
    L.ZSTD_compress.restype = C.c_size_t
    L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.ZSTD_isError.argtypes = [C.c_size_t]
    cap = L.ZSTD_compressBound(len(data))
    out = C.create_string_buffer(cap)
    n = L.ZSTD_compress(out, cap, data, len(data), level)
    assert not L.ZSTD_isError(n)
    return out.raw[:n]


def test_bzip2_xz_zstd_give_the_same_bytes(host_tests, tmp_path):
    """The other formats niffler sniffs (src/lib.rs:173-190): decoded by the system's libbz2 / liblzma / libzstd, bound
    at run time.  Files written by Python's bz2 / lzma modules and by libzstd: one stream, and two back to back."""
    import lzma
    rng = np.random.default_rng(79)
    data = fuzzgen.valid_file(rng, 6000, maxlen=150)
    want = "plain %d %016x" % (len(data), fnv1a(data))
    cut = len(data) // 3
    variants = {
        "a.bz2": bz2.compress(data),
        "multi.bz2": bz2.compress(data[:cut], 1) + bz2.compress(data[cut:]),
        "a.xz": lzma.compress(data, format=lzma.FORMAT_XZ),
        "multi.xz": lzma.compress(data[:cut], format=lzma.FORMAT_XZ, preset=1) + lzma.compress(data[cut:], format=lzma.FORMAT_XZ),
    }
    try:
        variants["a.zst"] = zstd_frame(data)
        variants["multi.zst"] = zstd_frame(data[:cut], 1) + zstd_frame(data[cut:], 19)
    except OSError:
        pass   # no libzstd.so.1 on this box to write test frames with
    for name, blob in variants.items():
        (tmp_path / name).write_bytes(blob)
        assert plain(host_tests, tmp_path / name) == (0, want), name
    for name in ("a.bz2", "a.xz", "a.zst"):
        if name in variants:
            (tmp_path / ("trunc." + name)).write_bytes(variants[name][: len(variants[name]) // 2])
            rc, out = plain(host_tests, tmp_path / ("trunc." + name))
            assert rc == 3 and out.startswith("error"), name
            (tmp_path / ("junk." + name)).write_bytes(variants[name][:64] + b"\x00junk" * 50)
            rc, out = plain(host_tests, tmp_path / ("junk." + name))
            assert rc == 3 and out.startswith("error"), name


def test_sniffing_errors(host_tests, tmp_path):
    data = b"@a\nACGT\n+\nIIII\n" * 100
    (tmp_path / "short").write_bytes(b"@a\nA")                      # niffler: FileTooShort
    rc, out = plain(host_tests, tmp_path / "short")
    assert rc == 3 and "less than five bytes" in out
    (tmp_path / "x.xz").write_bytes(bytes([0xfd, 0x37, 0x7a, 0x58, 0x5a, 0]) + b"junk")   # the magic, then nothing xz
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
