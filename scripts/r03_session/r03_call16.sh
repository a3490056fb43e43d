mkdir -p gpurun_out/r03
BENCH_SKIP_PCIE=1 timeout -k 10 300 python bench.py --strong-in-process --strong-first --no-cpu-baseline --no-ar-workloads --no-dominant > gpurun_out/r03/x.json 2> gpurun_out/r03/x.err
BENCH_SKIP_PCIE=1 timeout -k 10 300 python bench.py --strong-in-process --no-cpu-baseline --no-ar-workloads --no-dominant > gpurun_out/r03/bench_strong_y.json 2> gpurun_out/r03/bench_strong_y.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_strong_y.json') if l.startswith('{')][-1])
print('skip pcie, strong last:', 'value', round(d['value'],1), 'strong', round(d['strong']['value'],1), d['strong']['ms_per_step'])"
