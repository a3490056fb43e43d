mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r03/pytest42.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r03/pytest42.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python bench.py > gpurun_out/r03/bench_line2.json 2> gpurun_out/r03/bench_line2.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_line2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['metric','value','unit','ms_per_step','steps','warmup','n_gpus']})
print(d['roofline']['frac'], d['roofline'].get('end_to_end',{}).get('frac'))
print(d['strong_per_gpu_proxy']['value'], d['strong_per_gpu_proxy']['frac_of_value'])
print(d['ar_workloads']['checkerboard']['value'], d['ar_workloads']['basic']['value'])
PY
