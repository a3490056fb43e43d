mkdir -p gpurun_out/r03
timeout -k 10 500 python bench.py --gpus 2 --dist-backend gloo --same-device --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03/rehearsal_gpus2.json 2> gpurun_out/r03/rehearsal_gpus2.err; echo "rc $?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/rehearsal_gpus2.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['metric','value','n_gpus','steps','scaling','ms_per_step']}); print(d.get('strong'))
PY
tail -3 gpurun_out/r03/rehearsal_gpus2.err
