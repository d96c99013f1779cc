"""Holds fastq-rs_amd/rust/ffi.rs — the binding a maintainer of the `fastq` crate would add; there is no rustc in the image,
so it is never compiled — to include/fastq_hip.h mechanically: every function (name, argument count, every argument's and the
return value's type), every #[repr(C)] struct (field order, types, hence offsets and size, also against the ctypes structs the
GPU tests run through) and every constant.  Either file drifting fails here.  The seam the binding sits in
is RecordSetIter::next / RecordRefIter::advance calling IdxRecord::from_buffer, src/lib.rs:262, 373."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fastq_hip.h")
RUST = os.path.join(ROOT, "fastq-rs_amd", "rust", "ffi.rs")

SCALARS = {"uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "uint8_t": "u8", "int": "c_int", "float": "f32", "double": "f64",
           "void": "c_void", "char": "c_char", "fqh_status": "c_int"}
SIZES = {"u64": 8, "u32": 4, "i32": 4, "u8": 1, "c_int": 4, "f32": 4, "f64": 8}


def strip_c(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    return src


def c_type(t, fnptrs):
    """'const uint8_t *' -> '*const u8', 'fqh_ctx **' -> '*mut *mut fqh_ctx', 'fqh_read_fn' -> its Rust spelling."""
    t = t.strip()
    if t in fnptrs:
        return fnptrs[t]
    stars = t.count("*")
    const = bool(re.search(r"\bconst\b", t))
    base = re.sub(r"\bconst\b|\bstruct\b|\*", " ", t).split()
    assert len(base) == 1, t
    r = SCALARS.get(base[0], base[0])
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def c_arg(a, fnptrs):
    """One C parameter -> its canonical type (arrays decay to pointers)."""
    a = a.strip()
    m = re.match(r"^(.*?)(\w+)\s*\[\s*(\w*)\s*\]$", a)
    if m:
        return c_type(m.group(1) + " *", fnptrs)
    m = re.match(r"^(.*?)(\w+)$", a, flags=re.S)
    assert m, a
    return c_type(m.group(1), fnptrs)


def parse_header():
    src = strip_c(open(HEADER).read())
    consts = {}
    for name, val in re.findall(r"#define\s+(FQH_\w+)\s+([^\n]+)", src):
        val = val.strip().strip("()")
        m = re.fullmatch(r"(0x[0-9A-Fa-f]+|\d+)[uU]?", val)
        if m:
            consts[name] = int(m.group(1), 0)
        elif val == "UINT64_MAX":
            consts[name] = 2**64 - 1
        elif re.fullmatch(r"68u \* 1024u", val):
            consts[name] = 68 * 1024
    for body in re.findall(r"typedef\s+enum\s*\{(.*?)\}\s*fqh_status", src, flags=re.S):
        for name, val in re.findall(r"(FQH_\w+)\s*=\s*(\d+)", body):
            consts[name] = int(val)
    fnptrs = {}
    for ret, name, args in re.findall(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\(([^)]*)\)\s*;", src):
        a = [c_arg(x, {}) for x in args.split(",")]
        fnptrs[name] = 'extern "C" fn(%s) -> %s' % (", ".join(a), c_type(ret, {}))
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(fqh_\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"^((?:const\s+)?\w+(?:\s*\*)*)\s*(.*)$", decl, flags=re.S)
            base, names = m.group(1), m.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                stars = nm.count("*")
                nm = nm.replace("*", "").strip()
                arr = re.match(r"(\w+)\s*\[\s*(\d+)\s*\]", nm)
                t = c_type(base + " " + "*" * stars, fnptrs)
                if arr:
                    fields.append((arr.group(1), "[%s; %s]" % (t, arr.group(2))))
                else:
                    fields.append((nm, t))
        structs[name] = fields
    funcs = {}
    for ret, name, args in re.findall(r"\n\s*((?:const\s+)?[\w]+(?:\s*\*)*)\s*\b(fqh_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = args.strip()
        a = [] if args in ("", "void") else [c_arg(x, fnptrs) for x in args.split(",")]
        funcs[name] = (a, None if ret.strip() == "void" else c_type(ret, fnptrs))
    return consts, structs, funcs, fnptrs


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def norm(t):
    return re.sub(r"\s+", " ", t.strip())


def parse_rust():
    src = open(RUST).read()
    src = re.sub(r"//[^\n]*", "", src)
    consts = {}
    for name, val in re.findall(r"const\s+(FQH_\w+)\s*:\s*\w+\s*=\s*([^;]+);", src):
        val = val.strip().replace("_", "")
        if val == "u64::MAX":
            consts[name] = 2**64 - 1
        else:
            consts[name] = int(eval(val, {"__builtins__": {}}))   # "68 * 1024", "0xFFFFFFFE"
    structs = {}
    for name, body in re.findall(r"pub\s+struct\s+(fqh_\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for f in split_top(body):
            f = f.strip()
            if not f:
                continue
            m = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.+)$", f, flags=re.S)
            fields.append((m.group(1), norm(m.group(2))))
        structs[name] = fields
    funcs = {}
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    for m in re.finditer(r"pub\s+fn\s+(fqh_\w+)\s*\(", block):
        i, depth = m.end(), 1
        while depth:                      # (argument types hold parentheses of their own: extern "C" fn(..) -> ..)
            depth += {"(": 1, ")": -1}.get(block[i], 0)
            i += 1
        args = block[m.end(): i - 1]
        ret = re.match(r"\s*(?:->\s*([^;]+))?;", block[i:]).group(1)
        a = []
        for x in split_top(args):
            mm = re.match(r"\s*\w+\s*:\s*(.+)$", x.strip(), flags=re.S)
            a.append(norm(mm.group(1)))
        funcs[m.group(1)] = (a, norm(ret) if ret else None)
    return consts, structs, funcs


def layout(fields, structs):
    """-> (size, [(name, offset)]) under the C rules (natural alignment)."""
    off, align_max, offs = 0, 1, []
    for name, t in fields:
        m = re.match(r"\[(.+); (\d+)\]", t)
        n = int(m.group(2)) if m else 1
        base = m.group(1) if m else t
        size = 8 if base.startswith("*") else SIZES[base]
        off = (off + size - 1) // size * size
        offs.append((name, off))
        off += size * n
        align_max = max(align_max, size)
    return (off + align_max - 1) // align_max * align_max, offs


def test_every_function_is_bound_with_the_header_s_signature():
    hc, hs, hf, _ = parse_header()
    rc, rs, rf = parse_rust()
    assert len(hf) >= 69
    missing = sorted(set(hf) - set(rf))
    extra = sorted(set(rf) - set(hf))
    assert not missing, "declared in include/fastq_hip.h, not bound in rust/ffi.rs: %s" % missing
    assert not extra, "bound in rust/ffi.rs, not in the header: %s" % extra
    for name, (args, ret) in hf.items():
        rargs, rret = rf[name]
        assert len(args) == len(rargs), (name, args, rargs)
        assert [norm(a) for a in args] == rargs, (name, args, rargs)
        assert (ret and norm(ret)) == rret, (name, ret, rret)


def test_repr_c_structs_have_the_header_s_layout():
    hc, hs, hf, _ = parse_header()
    rc, rs, rf = parse_rust()
    opaque = {"fqh_ctx", "fqh_stream", "fqh_comm"}
    assert set(hs) == set(rs) - opaque, (sorted(hs), sorted(rs))
    for name, fields in hs.items():
        assert [(n, norm(t)) for n, t in fields] == rs[name], (name, fields, rs[name])
    # ... and the layouts are the ones the GPU tests run through (the ctypes structs of fastq-rs_amd/binding.py)
    import __graft_entry__ as g
    pkg = g.load_package()
    import importlib
    B = importlib.import_module("fastq_rs_amd.binding")
    twins = {"fqh_carry": B.Carry, "fqh_summary": B.Summary, "fqh_idx_record": B.IdxRecord, "fqh_timing": B.Timing, "fqh_chunk": B.Chunk,
             "fqh_shard_result": B.ShardResult, "fqh_stream_times": B.StreamTimes}
    assert set(twins) == set(hs)
    for name, T in twins.items():
        size, offs = layout(hs[name], hs)
        assert size == C.sizeof(T), (name, size, C.sizeof(T))
        assert [n for n, _ in offs] == [f[0] for f in T._fields_], name
        assert [o for _, o in offs] == [getattr(T, f[0]).offset for f in T._fields_], name


def test_constants_agree():
    hc, _, _, _ = parse_header()
    rc, _, _ = parse_rust()
    assert len(hc) >= 34
    assert set(hc) == set(rc), "only in the header: %s; only in ffi.rs: %s" % (sorted(set(hc) - set(rc)), sorted(set(rc) - set(hc)))
    assert hc == rc, {k: (hc[k], rc[k]) for k in hc if hc[k] != rc[k]}
    # (and the ctypes side)
    import __graft_entry__ as g
    pkg = g.load_package()
    for k, v in hc.items():
        short = k[4:]
        if hasattr(pkg, short):
            assert getattr(pkg, short) == v, k


def test_the_parsers_see_what_is_there():
    """The test is only as good as its parsers: a few signatures spelled out by hand."""
    _, hs, hf, fp = parse_header()
    assert hf["fqh_create"] == (["c_int", "*mut *mut fqh_ctx"], "c_int")
    assert hf["fqh_strerror"] == (["c_int"], "*const c_char")
    assert hf["fqh_destroy"] == (["*mut fqh_ctx"], None)
    assert hf["fqh_carry_combine"][0][4] == "*const u64"            # const uint64_t back_zero_carry[4]
    assert hf["fqh_placement"][0][2] == "*mut f32"                  # float ms[10]
    assert hf["fqh_shard_stream_run"][0][1] == 'extern "C" fn(*mut c_void, *mut u8, u64, u64) -> c_int'
    assert hs["fqh_carry"] == [("base_offset", "u64"), ("nl_count", "u64"), ("back", "[u64; 4]")]
    assert layout(hs["fqh_chunk"], hs)[0] == 104
