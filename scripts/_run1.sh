timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 400 python bench.py --cpu-sample 0 --no-input-stage --no-secondary --steps 50 --warmup 10 --steady 50 --full-step 8 > gpurun_out/b_f32.json 2>gpurun_out/b_err.txt
python - <<'PY'
import json
r=json.loads(open('gpurun_out/b_f32.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r.get('steady'), r['roofline']['kernel'], r['roofline']['frac'], r['full_step'])
PY
