#!/bin/bash
# SDWA-with-SGPR probe, then the decoder tests with the SDWA common path; on failure the same tests with -DWD_NO_SDWA
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chain
mkdir -p $O
cd $R
hipcc -O2 --offload-arch=gfx950 scripts/r04_sdwa_probe.hip -o /tmp/sdwa 2>/dev/null && /tmp/sdwa | tee $O/sdwa.txt
T="tests/test_gpu_ar_codecs.py tests/test_gpu_scanline.py tests/test_gpu_reference_kats.py"
timeout -k 10 600 python -m pytest $T -m gpu -x -q > $O/pytest_sdwa.log 2>&1; rc=$?
tail -3 $O/pytest_sdwa.log
if [ $rc -ne 0 ]; then
  echo "== rebuilding with -DWD_NO_SDWA"
  cd cbench_basic_amd/csrc && rm -f rans.o scanline.o && make FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -DWD_NO_SDWA" > $O/rebuild.log 2>&1 || { tail -5 $O/rebuild.log; exit 1; }
  cd $R
  timeout -k 10 600 python -m pytest $T -m gpu -x -q > $O/pytest_nosdwa.log 2>&1; rc=$?
  tail -3 $O/pytest_nosdwa.log
fi
exit $rc
