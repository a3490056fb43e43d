#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_traffic.json.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
The passes are run SEPARATELY (one counter each, with --kernel-trace only), as the MI355X guide prescribes.
"""
import csv, glob, json, sys, collections

TRANSFORM = ("conv_tap_mfma_kernel", "deconv5s2_cout3", "conv5x5_cin4_gdn_persistent_kernel")


def fold(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not name.startswith(TRANSFORM):
            continue
        per[name][0] += 1
        per[name][1] += float(r["Counter_Value"])
    return {k: dict(launches=v[0], kib_total=v[1]) for k, v in per.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    fs, ws = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    launches = sum(v["launches"] for v in fs.values())
    kib = sum(v["kib_total"] for v in fs.values()) + sum(v["kib_total"] for v in ws.values())
    doc = dict(
        command="rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dominant",
        note="FETCH_SIZE/WRITE_SIZE in KiB as reported.  The weight slabs and gamma are read with 16-byte LDS-DMA pieces "
             "(the case MI355X_MICROARCH.md says FETCH_SIZE under-reports by 2x) but they are L2 hits re-read by every "
             "workgroup, not HBM streams; the activation patches are 4-byte-per-lane gathers, outside that calibration. "
             "No x2 correction is applied; treat the figure as a lower bound on fetch bytes.",
        per_kernel=dict(FETCH_SIZE=fs, WRITE_SIZE=ws), transform_launches=launches,
        hbm_bytes_per_launch_avg=kib * 1024 / max(launches, 1))
    json.dump(doc, open(out, "w"), indent=1)
    print(out, "launches", launches, "avg MB/launch %.1f" % (doc["hbm_bytes_per_launch_avg"] / 1e6))


if __name__ == "__main__":
    main()
