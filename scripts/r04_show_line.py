#!/usr/bin/env python3
"""Print the main figures of a bench.py JSON line."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
r = d["roofline"]
print("headline", round(d["value"], 1), "Mpix/s", round(d["ms_per_step"], 2), "ms/step; workers", c.get("workers"), c.get("workers_probe_mpix_s"), "lanes", c.get("token_lanes"),
      "bytes match", c.get("bytes_match_single_stream"))
print("  roofline frac", round(r["frac"], 3), "pass_ms", round(r["pass_ms"], 2), "e2e", round(r["end_to_end"]["frac"], 3), "dominant", round((r.get("dominant") or {}).get("frac", 0), 3),
      "traffic", r.get("traffic"), (r.get("traffic_source") or "")[:60])
if "pcie_inclusive" in d:
    print("  pcie", round(d["pcie_inclusive"]["value"], 1))
sp = d.get("strong_per_gpu_proxy") or {}
print("  strong proxy", {k: sp.get(k) for k in ("value", "workers", "workers_probe_mpix_s", "frac_of_value", "ms_per_step", "bytes_match_single_stream", "error")})
for w, a in (d.get("ar_workloads") or {}).items():
    if "error" in a:
        print(" ", w, "ERROR", a)
        continue
    ro = a["roofline"]
    print(" ", w, round(a["value"], 1), "Mpix/s; bytes match", a["config"].get("bytes_match_single_stream"), "workers", a["config"].get("workers"), "frac", round(ro["frac"], 4),
          "occupied", ro.get("frac_of_occupied_units"), "traffic", ro.get("traffic"), "enc/dec ms", ro.get("encode_launch_ms"), ro.get("decode_launch_ms"))
    if "levels" in a:
        print("    levels", {k: (round(v["value"], 1), round(v["bpp"], 3), v["bytes_match_single_stream"]) for k, v in a["levels"].items()})
    if "kodak_batch1" in a:
        print("    kodak", a["kodak_batch1"].get("value"), a["kodak_batch1"].get("error"))
    print("    cpu", a.get("cpu_baseline"))
print("  cpu_baseline", d.get("cpu_baseline"))
