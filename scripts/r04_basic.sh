#!/bin/bash
# bench.py --workload basic with the batched scan-line kernel: workers sweep (through gpurun)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04basic
mkdir -p $O
cd $R
for w in ${WORKERS:-3 4 6 2 1}; do
  timeout -k 10 200 python bench.py --workload basic --workers $w --steps 12 --warmup 4 --no-cpu-baseline > $O/basic_w$w.json 2> $O/basic_w$w.err || { echo "w=$w failed"; tail -3 $O/basic_w$w.err; }
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/basic_w$w.json") if l.startswith("{")][-1])
    print("workers $w:", round(d["value"], 1), "Mpix/s", round(d["ms_per_step"], 2), "ms/step", "bpp", d["config"]["bpp"])
except Exception as e:
    print("workers $w: no line", e)
PY
done
