set -x
mkdir -p gpurun_out/r03
PROBE_PERM=1 timeout -k 10 300 python scripts/mconv_probe.py w4_perm_wide: w4_perm_narrow:BASIC_MCONV_DEBUG=8 w8_perm_wide:BASIC_MCONV_DMA_WAVES=8 w8_perm_narrow:BASIC_MCONV_DMA_WAVES=8,BASIC_MCONV_DEBUG=8 w4_nostage:BASIC_MCONV_DEBUG=1 w4_nostore:BASIC_MCONV_DEBUG=4 gather:BASIC_MCONV_KERNEL=gather w4_again: > gpurun_out/r03/mconv_probe7.txt 2>&1
timeout -k 10 300 python scripts/mconv_probe.py w4_rowmajor: w8_rowmajor:BASIC_MCONV_DMA_WAVES=8 >> gpurun_out/r03/mconv_probe7.txt 2>&1; cat gpurun_out/r03/mconv_probe7.txt
timeout -k 10 300 python scripts/scanline_probe.py > gpurun_out/r03/scanline_probe1.txt 2>&1; cat gpurun_out/r03/scanline_probe1.txt
timeout -k 10 400 python bench.py --steps 12 > gpurun_out/r03/bench_default2.json 2> gpurun_out/r03/bench_default2.err; tail -c 2500 gpurun_out/r03/bench_default2.json; tail -3 gpurun_out/r03/bench_default2.err
