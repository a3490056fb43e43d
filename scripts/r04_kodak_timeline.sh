#!/bin/bash
# one stream, Kodak-shaped BaSIC items at batch 1: kernel timeline of the last item (where the compress / decompress time goes)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kodak_tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt0 -o k --output-format csv -- python3 $R/tools/run_benchmark.py --warmup --codec basic --synthetic 6 --height 512 --width 768 --batch-size 1 --workers 0 --complexity-levels 0 --out $O/run > $O/run.json 2> $O/run.err || { tail -5 $O/run.err; exit 1; }
python3 $R/scripts/kodak_timeline.py /tmp/kt0 95 > $O/timeline.txt 2>&1
head -70 $O/timeline.txt
