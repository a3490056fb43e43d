set -x
mkdir -p gpurun_out/r03
for flags in "--strong-in-process --strong-first" "--strong-in-process" ; do
timeout -k 10 300 python bench.py $flags --no-cpu-baseline --no-ar-workloads --no-dominant > gpurun_out/r03/bench_strong_x.json 2> gpurun_out/r03/bench_strong_x.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_strong_x.json') if l.startswith('{')][-1])
print('$flags', 'value', round(d['value'],1), 'strong', round(d['strong']['value'],1), d['strong']['ms_per_step'], 'pcie', round(d['pcie_inclusive']['value'],1))"
done
