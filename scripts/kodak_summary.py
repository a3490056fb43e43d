#!/usr/bin/env python3
"""Summarise the JSON files written by scripts/kodak_shaped.sh: per (config, complexity level) the batch times and Mpix/s."""
import glob, json, os, sys
d0 = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for f in sorted(glob.glob(os.path.join(d0, "kodak_*.json"))):
    d = json.load(open(f))
    rows = {}
    for k, v in d.items():
        for t in ("time_compress", "time_decompress", "compression_ratio", "psnr"):
            if k.endswith(t):
                rows.setdefault(k[:-len(t)].rstrip("_"), {})[t] = v
    nb = 24 if "b24" in f else 1
    name = os.path.basename(f)[:-5]
    for lvl, r in rows.items():
        tot = r["time_compress"] + r["time_decompress"]
        print(f"{name:16s} {lvl or '-':9s} compress {r['time_compress']:8.1f} ms  decompress {r['time_decompress']:8.1f} ms  per batch of {nb:2d}: "
              f"{nb * 512 * 768 / tot / 1e3:7.1f} Mpix/s   bpp {r['compression_ratio'] * 96:.3f}  psnr {r['psnr']:.2f} dB")
