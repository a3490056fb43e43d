mkdir -p gpurun_out/r03
timeout -k 10 1000 python bench.py > gpurun_out/r03/bench_line.json 2> gpurun_out/r03/bench_line.err; echo "bench rc $?"; tail -c 3000 gpurun_out/r03/bench_line.json; tail -3 gpurun_out/r03/bench_line.err
