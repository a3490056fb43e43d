#!/bin/bash
# headline bench over worker counts / sharding / rANS packing (informational; writes gpurun_out/sweep.txt)
# BATCH=32 STEPS=96: one rank's share of the N = 8 strong-scaling leg (BASELINE configs[4]: 256 images over 8 GPUs)
# a 4th field of a SWEEP entry sets GPU_MAX_HW_QUEUES for that run
mkdir -p gpurun_out
out=${OUT:-gpurun_out/sweep.txt}
: > $out
SWEEP=${SWEEP:-2:steps:-1 3:steps:-1 4:steps:-1 2:images:-1 2:steps:1 2:steps:8}
for cfg in $SWEEP; do
  IFS=: read w by rw hq <<< "$cfg"
  echo "== batch ${BATCH:-256} workers $w shard-by $by rans-waves $rw hw-queues ${hq:-default}" >> $out
  if [ -n "$hq" ]; then export GPU_MAX_HW_QUEUES=$hq; else unset GPU_MAX_HW_QUEUES; fi
  timeout -k 10 200 python bench.py --batch ${BATCH:-256} --workers $w --shard-by $by --rans-waves $rw --steps ${STEPS:-12} --no-cpu-baseline --no-extra-legs --no-dominant 2>>gpurun_out/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['end_to_end']['frac'],3), 'call', round(d['config'].get('call_latency_ms',0),1))" >> $out || exit 1
done
cat $out
