"""CPU-side checks of the C-ABI boundary: libfastq_hip.so loads, exports every symbol
include/fastq_hip.h declares, reports the reference's exact error strings, and fails loudly
(no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fastq_hip.h")


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    p = g.load_package()
    if not os.path.exists(p.LIB_PATH):
        g.build()
    return p


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fqh_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(pkg):
    L = pkg.lib()
    decl = declared_symbols()
    assert len(decl) >= 20
    missing = [s for s in decl if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(pkg.EXPORTS) == decl  # the ctypes stub binds exactly what the header declares


def test_dynamic_symbol_table_is_the_header(pkg):
    """-fvisibility=hidden + csrc/exports.map: a host can link against what include/fastq_hip.h declares and nothing else — no
    fqh_internal_*, no fqh_ctx / fqh_stream members, no kernel handles."""
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert syms == declared_symbols(), sorted(set(syms) ^ set(declared_symbols()))
    assert len(syms) >= 65


def test_no_oracle_in_product(pkg):
    """The product library must not link or reference anything under oracle/."""
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "fqref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "fastq-rs_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".hpp", ".cpp", ".py")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "fqref" not in txt and "oracle/" not in txt and "import oracle" not in txt \
                    and "from oracle" not in txt, f


def test_error_strings_match_reference(pkg):
    # src/records.rs:143-146, 157-160, 234-237; src/lib.rs:279-282, 287-290
    assert pkg.strerror(pkg.E_HEADER) == "Fastq headers must start with '@'"
    assert pkg.strerror(pkg.E_SEP) == "Sequence and quality not separated by +"
    assert pkg.strerror(pkg.E_LEN_MISMATCH) == "Sequence and quality length mismatch"
    assert pkg.strerror(pkg.E_TRUNCATED) == "Possibly truncated input file"
    assert pkg.strerror(pkg.E_TOO_LONG) == "Fastq record is too long"


def test_struct_layouts(pkg):
    assert C.sizeof(pkg.Carry) == 48
    assert C.sizeof(pkg.Summary) == 72
    assert C.sizeof(pkg.IdxRecord) == 24
    assert C.sizeof(pkg.Timing) == 20
    assert pkg.lib().fqh_abi_version() == 1
    assert pkg.BUFSIZE == 69632


def test_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.FqhError) as e:
        pkg.Ctx(0)
    assert e.value.status == pkg.E_DEVICE


def test_missing_library_is_an_error(pkg, tmp_path, monkeypatch):
    import importlib
    b = importlib.import_module("fastq_rs_amd.binding")
    monkeypatch.setattr(b, "_LIB", None)
    monkeypatch.setattr(b, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        b.lib()
