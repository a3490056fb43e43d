mkdir -p gpurun_out/r03
for w in 5 6 6 4; do
  timeout -k 10 300 python bench.py --workload checkerboard --workers $w --no-cpu-baseline > gpurun_out/r03/bench_cb_w$w.json 2> gpurun_out/r03/bench_cb_w$w.err || echo "cb w$w failed"
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r03/bench_cb_w$w.json').read().strip().splitlines()[-1]); print('checkerboard workers $w:', round(d['value'],1), d['unit'], round(d['ms_per_step'],2))"
done
