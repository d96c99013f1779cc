#!/usr/bin/env python3
"""Summarises tools/prof.sh output into profiles/<tag>_*: the rocprofv3 --stats table, per-dispatch durations of the
kernels of the hot path (avg / min / max over the launches at the benchmark's size), PMC counters per launch, the HBM
traffic derived from FETCH_SIZE / WRITE_SIZE (FETCH_SIZE doubled for the 16-byte-per-lane streams as MI355X_MICROARCH.md
prescribes, calibrated in the same profile on the plain read kernel k_read_ceiling), and the agreement check between
rocprofv3's durations and the HIP-event figure bench.py printed in the traced run."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "round4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)
HOT = ("k_index_fast", "k_scan_stats", "k_emit_fast", "k_prefix", "k_finalize", "k_stats_commit", "k_stats_declined", "k_stats_edge", "k_stats_long_reduce", "k_stats_long", "k_long_census", "k_long_plan", "k_long_lists", "k_index_t", "k_stats_oct",
       "k_emit(", "k_read_ceiling", "k_stats_reduce")
txt = []


def short(name):
    for h in HOT:
        if h in name:
            return h.rstrip("(")
    return None


bench = None
log = os.path.join(out, "trace.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        bench = json.loads(lines[-1])
        with open(os.path.join(PROF, tag + "_bench_traced.json"), "w") as f:
            f.write(lines[-1])
res = {"source": "tools/prof.sh (rocprofv3 --kernel-trace --stats, then --pmc passes) over: python bench.py --steps 5 --warmup 2 "
                 "--no-cpu-baseline",
       "workload_bytes": bench["config"]["bytes_per_gpu"] if bench else None}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(PROF, tag + "_kernel_stats.csv"))
    txt.append("== rocprofv3 --kernel-trace --stats (%s)" % res["source"])
    for row in csv.DictReader(open(f)):
        txt.append("  %-64s calls %5s avg %12s ns min %12s max %12s total %14s ns %6s%%" % (
            row.get("Name", "")[:64], row.get("Calls"), row.get("AverageNs"), row.get("MinNs"), row.get("MaxNs"),
            row.get("TotalDurationNs"), row.get("Percentage")))
# per-dispatch durations; the launches at full size are the long ones (the smoke-sized ones of the same kernel are dropped)
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in dur.items():
    big = [x for x in d if x > 0.5 * max(d)]
    res[k] = {"calls": len(big), "avg_ms": round(sum(big) / len(big), 4), "min_ms": round(min(big), 4), "max_ms": round(max(big), 4)}
txt.append("== per-dispatch durations, launches at full size (ms): " + json.dumps({k: res[k] for k in dur}))
pmc = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            if k:
                pmc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
for k, cs in pmc.items():
    per = {}
    for c, v in cs.items():
        big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
        per[c] = sum(big) / len(big)
    res.setdefault(k, {})["pmc_per_launch"] = {c: round(x, 1) for c, x in sorted(per.items())}
    txt.append("== pmc per launch, %s: %s" % (k, json.dumps(res[k]["pmc_per_launch"])))
# HBM traffic: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts half of a 16-byte-per-lane stream (the guide's gfx950 note)
cal = None
if "k_read_ceiling" in res and res.get("workload_bytes") and "FETCH_SIZE" in res["k_read_ceiling"].get("pmc_per_launch", {}):
    cal = res["workload_bytes"] / (res["k_read_ceiling"]["pmc_per_launch"]["FETCH_SIZE"] * 1024.0)
    res["fetch_size_calibration"] = {"kernel": "k_read_ceiling", "bytes_read": res["workload_bytes"],
                                     "FETCH_SIZE_KiB": res["k_read_ceiling"]["pmc_per_launch"]["FETCH_SIZE"],
                                     "bytes_per_counted_byte": round(cal, 4)}
for k in ("k_index_fast", "k_scan_stats", "k_index_t", "k_stats_oct", "k_stats_long"):
    p = res.get(k, {}).get("pmc_per_launch", {})
    if "FETCH_SIZE" in p:
        rd = p["FETCH_SIZE"] * 1024.0 * (cal if cal else 2.0)
        wr = p.get("WRITE_SIZE", 0.0) * 1024.0
        res[k]["hbm_bytes_per_launch"] = int(rd + wr)
        res[k]["hbm_read_bytes"] = int(rd)
        res[k]["hbm_write_bytes"] = int(wr)
        if res.get("workload_bytes"):
            res[k]["traffic_over_algorithmic"] = round((rd + wr) / res["workload_bytes"], 4)
if bench:
    name = bench["roofline"]["kernel"]
    txt.append("== agreement check (same process): %s rocprofv3 avg %.4f ms (min %.4f, max %.4f) vs bench.py HIP events %.4f ms"
               % (name, res[name]["avg_ms"], res[name]["min_ms"], res[name]["max_ms"], bench["roofline"]["kernel_ms"]))
    if "stats" in bench and "k_scan_stats" in res:
        txt.append("== k_scan_stats rocprofv3 avg %.4f ms vs bench.py stats.kernel_ms %.4f ms" % (res["k_scan_stats"]["avg_ms"],
                                                                                         bench["stats"]["kernel_ms"]))
# the streamed leg: kernels and copies of `bench.py --stream-gib 32 --producer-threads 8` (rocprofv3 --kernel-trace --memory-copy-trace)
sd = os.path.join(out, "stream")
if os.path.isdir(sd):
    st = {}
    for f in glob.glob(os.path.join(sd, "**", "*memory_copy_trace.csv"), recursive=True):
        n = 0
        tot = 0.0
        nbytes = 0
        t0 = t1 = None
        for r in csv.DictReader(open(f)):
            if "HOST_TO_DEVICE" not in r.get("Direction", "").upper().replace(" ", "_"):
                continue
            b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if e - b < 1000000:   # the ring's slot copies take milliseconds; the rest (words, tails) does not count here
                continue
            n += 1
            tot += (e - b) / 1e6
            t0 = b if t0 is None else min(t0, b)
            t1 = e if t1 is None else max(t1, e)
        if n:
            st["h2d_slot_copies"] = n
            st["h2d_busy_ms"] = round(tot, 2)
            st["h2d_first_to_last_ms"] = round((t1 - t0) / 1e6, 2)
            st["h2d_busy_frac"] = round(tot / ((t1 - t0) / 1e6), 4)
    for f in glob.glob(os.path.join(sd, "**", "*kernel_trace.csv"), recursive=True):
        kt = defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                kt[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        st["kernel_busy_ms"] = {k: round(v, 2) for k, v in sorted(kt.items())}
    sl = os.path.join(out, "stream.log")
    if os.path.exists(sl):
        lines = [l for l in open(sl) if l.startswith("{")]
        if lines:
            st["bench_line"] = json.loads(lines[-1])
    res["stream_trace"] = st
    txt.append("== configs[3] streamed leg under rocprofv3 (--kernel-trace --memory-copy-trace): " + json.dumps({k: v for k, v in st.items() if k != "bench_line"}))
with open(os.path.join(PROF, tag + "_rocprof.json"), "w") as f:
    json.dump(res, f, indent=1, sort_keys=True)
with open(os.path.join(PROF, tag + "_rocprofv3_summary.txt"), "w") as f:
    f.write("\n".join(txt) + "\n")
print("\n".join(txt))
