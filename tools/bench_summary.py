"""tools/bench_summary.py FILE — the headline fields of a bench.py JSON line, for a quick look."""
import json
import sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.1f GB/s  ms_per_step %.4f  frac_step %.4f" % (j["value"], j["ms_per_step"], j["hbm_roofline_frac_whole_step"]))
print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "kernel_ms_min", "kernel_ms_max", "traffic")})
print("input_placement", j["config"].get("input_placement"))
print("best_allocation", j.get("best_allocation"))
print("read_ceiling", j.get("read_ceiling_gbs"), j.get("read_ceiling_ms"))
st = j.get("stats", {})
print("stats", {k: st[k] for k in st if k in ("ms", "gbs", "hbm_frac", "kernel_ms", "route")} or list(st)[:12])
print("cpu", {k: v for k, v in j.get("cpu_baseline", {}).items() if k != "sample"})
for k, v in (j.get("host_api") or {}).items():
    if isinstance(v, dict):
        print("host_api %-28s" % k, v.get("gbs"), v.get("seconds_warm"), v.get("error", ""))
ss = j.get("sharded_stream")
if ss:
    for n in ("producer", "registered", "pinned_replay"):
        print("sharded", n, ss[n]["seconds"], ss[n]["gbs_aggregate"], ss[n].get("ratio_vs_n1"))
    print("ring_setup", ss["ring_setup_seconds"], "bytes_per_gpu", ss["bytes_per_gpu"])
if "stream" in j:
    print("stream", j["stream"]["gbs"])
