#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_traffic.json.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [git-sha]
The passes are run SEPARATELY (one counter each, with --kernel-trace only), as the MI355X guide prescribes.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly HALF the bytes of a streaming read.  The guide
states it for 16 B per lane; scripts/micro/fetch_calib.hip (output: profiles/r02_fetch_calibration.txt) shows the same
factor 0.500 on this part for EVERY access shape the transform kernels use -- global_load_dwordx4, global_load_lds with 16 B
and with 4 B per lane, and the row-wise patch gathers -- so the fetch of every kernel is doubled.  WRITE_SIZE is exact for
the store shapes used.
"""
import csv, glob, json, sys, collections

TRANSFORM = ("conv_tap_mfma_kernel", "deconv5s2_cout3", "conv5x5_cin4_gdn_persistent_kernel")
FETCH_X2 = TRANSFORM   # every kernel: see the calibration above


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
    git = sys.argv[4] if len(sys.argv) > 4 else "?"
    fs, ws = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    for k, v in fs.items():
        v["x2_applied"] = k.startswith(FETCH_X2)
        v["kib_corrected"] = v["kib_total"] * (2.0 if v["x2_applied"] else 1.0)
    launches = sum(v["launches"] for v in fs.values())
    raw = sum(v["kib_total"] for v in fs.values()) + sum(v["kib_total"] for v in ws.values())
    kib = sum(v["kib_corrected"] for v in fs.values()) + sum(v["kib_total"] for v in ws.values())
    doc = dict(
        command="rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --workers 1 --steps 2 --warmup 1 --no-cpu-baseline --no-dominant --no-extra-legs",
        git=git,
        note="FETCH_SIZE x2 for every kernel (calibrated: profiles/r02_fetch_calibration.txt); WRITE_SIZE as reported; FETCH counts L2 -> fabric requests, Infinity-Cache hits included",
        per_kernel=dict(FETCH_SIZE=fs, WRITE_SIZE=ws), transform_launches=launches,
        hbm_bytes_per_launch_uncorrected=raw * 1024 / max(launches, 1),
        hbm_bytes_per_launch_avg=kib * 1024 / max(launches, 1))
    json.dump(doc, open(out, "w"), indent=1)
    print(out, "launches", launches, "avg MB/launch %.1f (uncorrected %.1f)" % (doc["hbm_bytes_per_launch_avg"] / 1e6, doc["hbm_bytes_per_launch_uncorrected"] / 1e6))


if __name__ == "__main__":
    main()
