#!/bin/bash
# BASELINE configs[1..3] on 24 synthetic Kodak-shaped images (3x512x768; real Kodak is not available offline):
# the reference's own artefacts (metrics.csv / metrics_2d.csv) land under gpurun_out/kodak_*; the printed JSON is the summary.
set -e
out=${1:-gpurun_out}
python tools/run_benchmark.py --warmup --codec hyperprior --synthetic 24 --height 512 --width 768 --batch-size 1  --out $out/kodak_hp_b1   > $out/kodak_hp_b1.json
python tools/run_benchmark.py --warmup --codec hyperprior --synthetic 24 --height 512 --width 768 --batch-size 24 --out $out/kodak_hp_b24  > $out/kodak_hp_b24.json
python tools/run_benchmark.py --warmup --codec topogroup --method checkerboard --synthetic 24 --height 512 --width 768 --batch-size 24 --out $out/kodak_ckbd_b24 > $out/kodak_ckbd_b24.json
python tools/run_benchmark.py --warmup --codec basic --synthetic 24 --height 512 --width 768 --batch-size 1 --complexity-levels 0 1 2 3 4 5 6 7 --out $out/kodak_basic_b1 > $out/kodak_basic_b1.json
