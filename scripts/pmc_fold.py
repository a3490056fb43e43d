#!/usr/bin/env python3
"""Generic fold of rocprofv3 --pmc passes: per kernel (name prefix filter) the per-launch average of every counter found.
usage: pmc_fold.py <out.json> <prefix,prefix,...> <pass_dir> [<pass_dir> ...]"""
import collections, csv, glob, json, sys


def main():
    out, keep, dirs = sys.argv[1], tuple(sys.argv[2].split(",")), sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if not name.startswith(keep):
                    continue
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[name][r["Counter_Name"]] += 1
    doc = {name: {k: v / cnt[name][k] for k, v in c.items()} | {"launches": max(cnt[name].values())} for name, c in acc.items()}
    json.dump(doc, open(out, "w"), indent=1)
    for name, c in doc.items():
        print(name)
        for k, v in sorted(c.items()):
            print(f"    {k:32s} {v:16.1f}")


if __name__ == "__main__":
    main()
