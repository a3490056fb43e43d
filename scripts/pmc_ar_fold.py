#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of the AR workloads (TCC_EA0 request counters, separate passes) into profiles/r04_pmc_ar.json.

usage: pmc_ar_fold.py <out.json> <git-sha> <workload>=<dir>[,<dir>...] ...
Per workload: the dominant kernels' L2 -> fabric traffic per launch from the request COUNTS (FETCH_SIZE / WRITE_SIZE abort on these
kernels, the request counters they are derived from do not):
    read  bytes = 32 B x RDREQ_32B + 64 B x (RDREQ - RDREQ_32B), the 64-byte share DOUBLED (MI355X_MICROARCH.md, HBM section, and
                  profiles/r02_fetch_calibration.txt: gfx950 tallies a 128-byte read request as 64 bytes)
    write bytes = 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B)
Infinity-Cache hits are included (the counters sit on the L2's memory side)."""
import collections, csv, glob, json, sys

KEEP = {"basic": ("scanline_batched_kernel",), "checkerboard": ("masked_conv_",)}


def fold(dirs, keep):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if name.startswith(keep):
                    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[name][r["Counter_Name"]] += 1
    return acc, cnt


def main():
    out, git = sys.argv[1], sys.argv[2]
    doc = dict(git=git, note="TCC_EA0_RDREQ / RDREQ_32B / WRREQ / WRREQ_64B request counts x request size, 64-byte reads x2 (gfx950 tallies 128-byte "
                             "requests as 64 bytes: profiles/r02_fetch_calibration.txt); L2 -> fabric, Infinity-Cache hits included",
               command="rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --workload W --workers 1 --steps 1 --warmup 1 --levels 0 --no-kodak-leg --no-cpu-baseline",
               workloads={})
    for spec in sys.argv[3:]:
        wl, dirs = spec.split("=")
        acc, cnt = fold(dirs.split(","), KEEP[wl])
        kernels, tot_bytes, tot_launches = {}, 0.0, 0
        for name, c in acc.items():
            n = max(cnt[name].values())
            rd, rd32 = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            wr, wr64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            rd_b = 32.0 * rd32 + 2 * 64.0 * (rd - rd32)
            wr_b = 64.0 * wr64 + 32.0 * (wr - wr64)
            kernels[name] = dict(launches=n, rdreq=rd / n, rdreq_32b=rd32 / n, wrreq=wr / n, wrreq_64b=wr64 / n,
                                 read_bytes_per_launch=rd_b / n, write_bytes_per_launch=wr_b / n, counters=sorted(c))
            tot_bytes += rd_b + wr_b
            tot_launches += n
        doc["workloads"][wl] = dict(kernels=kernels, launches=tot_launches, hbm_bytes_per_launch=tot_bytes / max(tot_launches, 1))
        print(wl, "launches", tot_launches, "MB per launch %.2f" % (tot_bytes / max(tot_launches, 1) / 1e6))
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
