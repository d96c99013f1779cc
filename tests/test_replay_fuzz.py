"""csrc/replay.h (the host-side replay of the reference's Buffer: "Fastq record is too long", RecordSet cuts) against the
oracle on thousands of random inputs at the reference's fuzzing BUFSIZE of 64 (src/lib.rs:126-127), built with
AddressSanitizer + UBSan (the reference relies on Rust's ownership checks; the C++ host code gets
sanitizers instead).  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replay_differential_fuzz_under_asan_ubsan(tmp_path, fqref):
    exe = str(tmp_path / "replay_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "replay_fuzz.cpp"), "-o", exe, "-L", os.path.join(ROOT, "oracle"),
                           "-lfqref", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.run([exe, "3000", "7"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout
    assert int(out.stdout.split(" too-long cases")[0].split(", ")[-1]) > 100
