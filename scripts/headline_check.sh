#!/bin/bash
# Full GPU suite, then the fp32 headline with its roofline object and the fp32 training step (through gpurun).
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4
python3 bench.py --steps 50 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 8 2>/dev/null > gpurun_out/headline_check.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/headline_check.json'))
r = d['roofline']
print('fp32 fwd+loss ms', d['ms_per_step'], 'value', d['value'], 'loss', d['loss'])
print('roofline', {k: r.get(k) for k in ('kernel', 'achieved', 'peak', 'frac', 'avg_launch_us', 'fp32_equivalent_tflops', 'fp32_equivalent_over_fp32_mfma_peak', 'traffic')})
print('full step ms', d['full_step']['ms_per_step'])
PY
