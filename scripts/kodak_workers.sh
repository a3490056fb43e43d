#!/bin/bash
# the reference's own test setting (24 Kodak-shaped images, batch 1) through the harness with concurrent stream workers
out=${1:-gpurun_out}/kodak_workers
mkdir -p $out
: > $out/summary.txt
for cfg in ${KODAK_CFGS:-hyperprior:0 hyperprior:3 hyperprior:6 basic:0 basic:3 basic:6}; do
  set -- ${cfg/:/ }
  lv=""; [ "$1" = basic ] && lv="--complexity-levels 0"
  timeout -k 10 280 python tools/run_benchmark.py --warmup --codec $1 --synthetic 24 --height 512 --width 768 --batch-size 1 --workers $2 $lv --out $out/$1_w$2 > $out/$1_w$2.json 2> $out/$1_w$2.err || { echo "$1 w$2 FAILED" >> $out/summary.txt; tail -3 $out/$1_w$2.err >> $out/summary.txt; continue; }
  python - "$1" "$2" $out/$1_w$2.json >> $out/summary.txt <<'PY'
import json, sys
m = json.load(open(sys.argv[3]))
wall = [v for k, v in m.items() if k.endswith("time_wall_dataset")][0]
tc = [v for k, v in m.items() if k.endswith("time_compress")][0]
td = [v for k, v in m.items() if k.endswith("time_decompress")][0]
print(f"{sys.argv[1]:10s} workers {sys.argv[2]}: dataset wall {wall:8.1f} ms = {24 * 512 * 768 / wall / 1e3:6.2f} Mpix/s   (per item: compress {tc:.1f} ms, decompress {td:.1f} ms)")
PY
done
cat $out/summary.txt
