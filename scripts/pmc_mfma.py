#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (one directory per pass, --kernel-trace only) into profiles/rNN_pmc_mfma.json:
per transform kernel the MFMA busy fraction, the fp32 MFMA operation count and the LDS bank-conflict share.

usage: pmc_mfma.py <out.json> <pass_dir> [<pass_dir> ...]
Counters used when present: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_F32,
SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE (MI355X_MICROARCH.md: conflict = extra LDS cycles, idx_active = all LDS cycles).
"""
import collections, csv, glob, json, sys

KEEP = ("conv_tap_mfma_kernel", "deconv5s2_cout3", "conv5x5_cin4_gdn_persistent_kernel", "masked_conv", "rans_")


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if not name.startswith(KEEP):
                    continue
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[name][r["Counter_Name"]] += 1
    doc = dict(command="rocprofv3 --kernel-trace --pmc <counters> (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dominant",
               note="sums over all launches of a kernel in the run; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES-or-GRBM_GUI_ACTIVE x CUs "
                    "as reported by the tool's own MfmaUtil expression when that derived metric was collected); lds_conflict = "
                    "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
               kernels={})
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)):
        e = dict(launches=max(launches[name].values()))
        for k, v in c.items():
            e[k] = v
        if "MfmaUtil" in c:
            e["mfma_util_percent_avg"] = c["MfmaUtil"] / launches[name]["MfmaUtil"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in c:
            e["mfma_f32_flops"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
        doc["kernels"][name] = e
    json.dump(doc, open(out, "w"), indent=1)
    for name, e in doc["kernels"].items():
        print(f"{name[:52]:52s} launches {e['launches']:4d}  MfmaUtil {e.get('mfma_util_percent_avg', float('nan')):6.1f} %  "
              f"LDS conflict share {100 * e.get('lds_conflict_share', float('nan')):5.1f} %  MFMA GFLOP {e.get('mfma_f32_flops', 0) / 1e9:10.1f}")


if __name__ == "__main__":
    main()
