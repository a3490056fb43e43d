mkdir -p gpurun_out/r03
timeout -k 10 1000 python bench.py > gpurun_out/r03/bench_line3.json 2> gpurun_out/r03/bench_line3.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_line3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['metric','value','unit','ms_per_step','steps','warmup','n_gpus']})
print(d['roofline']['frac'], d['roofline'].get('end_to_end',{}).get('frac'), d['config'].get('workers'), d.get('pcie_inclusive'))
print(d['strong_per_gpu_proxy']['value'], d['strong_per_gpu_proxy']['frac_of_value'])
print(d['ar_workloads']['checkerboard']['value'], d['ar_workloads']['basic']['value'])
PY
timeout -k 10 500 python bench.py --gpus 2 --dist-backend gloo --same-device --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03/rehearsal_gpus2.json 2> gpurun_out/r03/rehearsal_gpus2.err; echo "rehearsal rc $?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/rehearsal_gpus2.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','n_gpus','steps','scaling','ms_per_step']}); print(d.get('strong'))
PY
