"""tools/isa_report.py FILE.s KERNEL_SUBSTRING [--blocks] — static look at one kernel of a `hipcc -S --cuda-device-only` dump:
registers / spills / scratch, instruction classes per loop depth, and (with --blocks) every basic block with its loop
header, so that a change to a hot loop can be judged before it costs GPU time (vector instructions are what bounds
k_scan_stats).  Scratch accesses and `s_waitcnt vmcnt` inside loops are listed: both have cost a
millisecond before."""
import collections
import re
import sys


def cat(op):
    if op.startswith("v_"):
        return "V"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64"):
        return "B"
    if op.startswith("s_waitcnt"):
        return "W"
    if op.startswith("s_nop"):
        return "N"
    if op.startswith("s_"):
        return "S"
    if op.startswith("ds_"):
        return "L"
    if op.startswith("scratch_"):
        return "X"
    return "M"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    src = open(path).read()
    names = re.findall(r"^(_Z\w+):\s*(?:;.*)?$", src, re.M)
    name = [n for n in names if key in n]
    if not name:
        sys.exit("no kernel matches %r in %s" % (key, names))
    name = name[0]
    body = re.search(r"^%s:(.*?)^\s+s_endpgm" % re.escape(name), src, re.S | re.M).group(1)
    meta = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), src, re.S)
    print(name)
    if meta:
        for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
            m = re.search(r"\.%s:\s+(\d+)" % k, src[src.find(".name:           " + name) - 1500: src.find(".name:           " + name) + 1500])
            if m:
                print("  %-28s %s" % (k, m.group(1)))
    blocks = []
    cur = {"name": "entry", "depth": 0, "hdr": "", "ins": []}
    pending_label = None
    for line in body.split("\n"):
        s = line.strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", s)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "depth": 0, "hdr": m.group(2) or "", "ins": []}
            d = re.search(r"Depth=(\d+)", cur["hdr"])
            if d:
                cur["depth"] = int(d.group(1))
            continue
        if s.startswith(";"):
            d = re.search(r"Depth=(\d+)", s)
            if d and not cur["ins"]:
                cur["depth"] = max(cur["depth"], int(d.group(1)))
                cur["hdr"] += " " + s
            continue
        if s.startswith((".", "/")):
            continue
        cur["ins"].append(s.split(";")[0].strip())
    blocks.append(cur)
    per_depth = collections.defaultdict(collections.Counter)
    for b in blocks:
        c = collections.Counter(cat(i.split()[0]) for i in b["ins"])
        b["c"] = c
        per_depth[b["depth"]].update(c)
    print("  per loop depth (static): V vector, S scalar, B branch, L LDS, M vmem, X scratch, W waitcnt")
    for d in sorted(per_depth):
        c = per_depth[d]
        print("    depth %d: V=%4d S=%4d B=%3d L=%3d M=%3d X=%2d W=%3d" % (d, c["V"], c["S"], c["B"], c["L"], c["M"], c["X"], c["W"]))
    tot = collections.Counter()
    for d in per_depth:
        tot.update(per_depth[d])
    print("    total  : V=%4d S=%4d B=%3d L=%3d M=%3d X=%2d W=%3d" % (tot["V"], tot["S"], tot["B"], tot["L"], tot["M"], tot["X"], tot["W"]))
    ops = collections.Counter(i.split()[0] for b in blocks if b["depth"] >= 2 for i in b["ins"])
    print("  most frequent in depth >= 2:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
    for b in blocks:
        if b["depth"] >= 1:
            for i in b["ins"]:
                if i.startswith("scratch_") or (i.startswith("s_waitcnt") and "vmcnt" in i):
                    print("  in loop (depth %d, %s): %s" % (b["depth"], b["name"], i))
    if show_blocks:
        for b in blocks:
            c = b["c"]
            at = sum(1 for i in b["ins"] if i.startswith(("ds_sub", "ds_add")))
            hdr = re.sub(r"\s+", " ", b["hdr"])[:46]
            print("  %-10s d%d n=%4d V=%3d S=%3d L=%2d at=%2d M=%2d X=%d B=%d  %s | %s" % (
                b["name"], b["depth"], len(b["ins"]), c["V"], c["S"], c["L"], at, c["M"], c["X"], c["B"], hdr,
                b["ins"][-1] if b["ins"] else ""))


if __name__ == "__main__":
    main()
