#!/bin/bash
# round-2 evidence for the AR workloads (run on the GPU box through gpurun); outputs under gpurun_out/r02prof_ar/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof_ar
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in checkerboard basic; do
  rocprofv3 --kernel-trace --stats -d $O/t_$wl -o t --output-format csv -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/trace_$wl.log 2>&1
  cp $(find $O/t_$wl -name "*kernel_stats.csv" | head -1) $O/r02_ar_kernel_stats_$wl.csv
  rm -rf $O/t_$wl
  rocprofv3 --kernel-trace --pmc MfmaUtil -d $O/m_$wl -o m --output-format csv -- python3 $R/bench.py --workload $wl --workers 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/mfma_$wl.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/l_$wl -o l --output-format csv -- python3 $R/bench.py --workload $wl --workers 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/lds_$wl.log 2>&1
  (cd $R && python scripts/pmc_mfma.py gpurun_out/r02prof_ar/r02_pmc_ar_$wl.json gpurun_out/r02prof_ar/m_$wl gpurun_out/r02prof_ar/l_$wl > gpurun_out/r02prof_ar/pmc_$wl.txt)
  rm -rf $O/m_$wl $O/l_$wl
  (cd $R && python bench.py --workload $wl > gpurun_out/r02prof_ar/r02_bench_$wl.json 2> gpurun_out/r02prof_ar/bench_$wl.err)
  echo $wl done
done
