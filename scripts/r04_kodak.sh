#!/bin/bash
# Kodak-shaped BaSIC items (batch 1) through the harness: pipelined kernel (default at batch 1) vs batched kernel (BASIC_SCAN_BATCHED_FROM=1)
cd $GRAFT_REPO_ROOT
for from in 3 1; do
  echo "== BASIC_SCAN_BATCHED_FROM=$from"
  BASIC_SCAN_BATCHED_FROM=$from KODAK_CFGS="${KODAK_CFGS:-basic:0 basic:3 basic:6}" bash scripts/kodak_workers.sh gpurun_out/r04kodak_$from 2>&1 | grep "workers"
done
