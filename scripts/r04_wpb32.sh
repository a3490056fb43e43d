#!/bin/bash
# 32 images per step (one GPU's share of the 8-GPU strong-scaling leg): image streams per rANS workgroup 8 (default) vs 4 vs 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wpb32
mkdir -p $O
cd $R
for w in 8 4 2 8 4; do
  timeout -k 10 200 python bench.py --batch 32 --steps 48 --warmup 6 --workers 6 --token-lanes 4 --rans-waves $w --no-cpu-baseline --no-extra-legs --no-ar-workloads --no-dominant > $O/w$w.json 2> $O/w$w.err || { tail -3 $O/w$w.err; continue; }
  python - $w $O/w$w.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("streams per rANS workgroup", sys.argv[1], ":", round(d["value"], 1), "Mpix/s,", round(d["ms_per_step"], 3), "ms per 32-image step, bytes match", d["config"].get("bytes_match_single_stream"))
PY
done
