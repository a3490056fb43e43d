mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PROBE_SHAPES=1
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/scan_trace -o st --output-format csv -- python3 scripts/scanline_probe.py > gpurun_out/r03/scan_trace.log 2>&1
f=$(find gpurun_out/r03/scan_trace -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -14
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03/scan_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last persistent encode launch and what surrounds it
idx = [i for i, r in enumerate(rows) if "scanline_persistent_kernel<false>" in r["Kernel_Name"]]
i = idx[-1]
t0 = int(rows[i - 3]["Start_Timestamp"])
for r in rows[i - 3:i + 8]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.3f} ms  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us  {r["Kernel_Name"][:80]}')
PY
rm -rf gpurun_out/r03/scan_trace
