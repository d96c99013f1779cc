#!/usr/bin/env python3
"""Writes tests/golden/reference_unit_tests.json.

The vectors are DATA transcribed from the reference crate's own unit tests
(/root/reference/src/lib.rs:611-811) and its one executing doc-test (src/lib.rs:474-508): the
input bytes each test feeds to `Parser::new(Cursor::new(..))` and the observable results the test
asserts.  No reference source text is kept here.  The reference is Rust and there is no
rustc/cargo in the image, so the vectors cannot be produced by *running* the reference; they are
what its tests pin, which is exactly what the oracle (oracle/fqref.c) must reproduce.

Large inputs are stored as recipes: a list of ["lit", text] and ["rep", text, count] parts.
`kind` tells the test which driver of the reference the vector exercises:
  each          Parser::each                (lib.rs:221)
  record_sets   Parser::record_sets         (lib.rs:428)
  parallel_each Parser::parallel_each(n,..) (lib.rs:509)
"""
import json
import os

BUFSIZE = 68 * 1024  # src/lib.rs:128-129

V = []


def vec(name, src, kind, parts, expect, note=""):
    V.append({"name": name, "src": src, "kind": kind, "input": parts, "expect": expect, "note": note})


rec = lambda h, s, q: {"head": h, "seq": s, "qual": q}

vec("correct", "src/lib.rs:616-652", "each",
    [["lit", "@hi\nNN\n+\n++\n@hallo\nTCC\n+\nabc\n"]],
    {"ok": True, "n_records": 2,
     "records": [rec("hi", "NN", "++"), rec("hallo", "TCC", "abc")],
     # OwnedRecord::write re-serialises to these bytes (12 and 17 bytes)
     "owned_write": ["@hi\nNN\n+\n++\n", "@hallo\nTCC\n+\nabc\n"]})

vec("empty_id", "src/lib.rs:654-666", "each",
    [["lit", "@\nNN\n+\n++\n"]],
    {"ok": True, "n_records": 1, "records": [rec("", "NN", "++")]})

vec("missing_lines", "src/lib.rs:668-686", "each",
    [["lit", "@hi\nNN\n+\n++\n@hi\nNN"]],
    {"ok": False, "error_kind": "InvalidData", "n_records": 1,
     "records": [rec("hi", "NN", "++")]},
    note="callback sees record 1, then Err(InvalidData)")

vec("truncated", "src/lib.rs:688-697", "each",
    [["lit", "@hi\nNN\n+\n++"]],
    {"ok": False, "n_records": 0})

vec("second_idline", "src/lib.rs:699-714", "each",
    [["lit", "@hi\nNN\n+hi\n++\n@hi\nNN\n+hi\n++\n"]],
    {"ok": True, "n_records": 2,
     "records": [rec("hi", "NN", "++"), rec("hi", "NN", "++")],
     # RefRecord::write dumps the raw 14 bytes including "+hi"
     "raw_write": ["@hi\nNN\n+hi\n++\n", "@hi\nNN\n+hi\n++\n"]})

vec("windows_lineend", "src/lib.rs:716-727", "each",
    [["lit", "@hi\r\nNN\r\n+\r\n++\r\n@hi\r\nNN\r\n+\r\n++\r\n"]],
    {"ok": True, "n_records": 2,
     "records": [rec("hi", "NN", "++"), rec("hi", "NN", "++")]})

vec("length_mismatch", "src/lib.rs:729-738", "each",
    [["lit", "@hi\nNN\n+\n+\n"]],
    {"ok": False, "n_records": 0})

vec("huge_incomplete", "src/lib.rs:740-750", "each",
    [["lit", "@"], ["rep", "longid", BUFSIZE]],
    {"ok": False})

vec("bufflen", "src/lib.rs:752-774", "parallel_each",
    [["lit", "@"], ["rep", "a", BUFSIZE - 8], ["lit", "\nA\n+\nB\n"]],
    {"ok": True, "n_threads": 2, "sum_of_worker_counts": 1},
    note="a record of exactly BUFSIZE bytes at file start is accepted")

vec("refset", "src/lib.rs:776-791", "record_sets",
    [["lit", "@hi\nNN\n+\n++\n@hi\nNN\n+\n++\n"]],
    {"ok": True, "total_records": 2,
     "records": [rec("hi", "NN", "++"), rec("hi", "NN", "++")]})

vec("refset_incomplete", "src/lib.rs:793-798", "record_sets",
    [["lit", "@hi\nNN\n+\n++\n@hi\nNN\n+\n++"]],
    {"ok": False})

vec("refset_huge_incomplete", "src/lib.rs:800-810", "record_sets",
    [["lit", "@"], ["rep", "longid", BUFSIZE]],
    {"ok": False})

vec("doctest_parallel_each", "src/lib.rs:474-508", "parallel_each",
    [["lit", "@hi\nATTAATTAATTA\n+\n++++++++++++\n"]],
    {"ok": True, "n_threads": 4, "sum_of_worker_counts": 1,
     "records": [rec("hi", "ATTAATTAATTA", "++++++++++++")]},
    note="some worker finds a record whose seq starts with ATTAATTA")

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_unit_tests.json")
    with open(out, "w") as f:
        json.dump({"bufsize": BUFSIZE, "vectors": V}, f, indent=1)
    print("wrote", out, len(V), "vectors")
