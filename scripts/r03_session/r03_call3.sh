set -x
mkdir -p gpurun_out/r03
timeout -k 10 300 python scripts/mconv_probe.py default: gather:BASIC_MCONV_KERNEL=gather nostage:BASIC_MCONV_DEBUG=1 nomfma:BASIC_MCONV_DEBUG=2 nostore:BASIC_MCONV_DEBUG=4 nomfma_nostore:BASIC_MCONV_DEBUG=6 nostage_nostore:BASIC_MCONV_DEBUG=5 > gpurun_out/r03/mconv_probe1.txt 2>&1; cat gpurun_out/r03/mconv_probe1.txt
timeout -k 10 900 python -m pytest tests/test_gpu_scanline.py tests/test_gpu_ar_codecs.py -q -m gpu -x > gpurun_out/r03/pytest3.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r03/pytest3.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/trace_b32 -- python bench.py --batch 32 --workers 4 --steps 48 --no-cpu-baseline --no-extra-legs --no-dominant > gpurun_out/r03/trace_b32.log 2>&1; tail -2 gpurun_out/r03/trace_b32.log | cut -c1-400
python scripts/timeline.py gpurun_out/r03/trace_b32 24 > gpurun_out/r03/timeline_b32.txt 2>&1; tail -8 gpurun_out/r03/timeline_b32.txt
find gpurun_out/r03/trace_b32 -name "*.csv" -size +20M -delete
