#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward or training" 2>&1 | tail -3
python3 bench.py --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 16 > gpurun_out/r04_b7.json 2> gpurun_out/r04_b7.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b7.json'))
print('f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'full', d['full_step']['ms_per_step'], 'bf16 full', d['bf16']['full_step']['ms_per_step'])
"
python3 bench.py --config gssd --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 16 > gpurun_out/r04_b8.json 2>> gpurun_out/r04_b7.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b8.json'))
print('gssd f32 ms', d['ms_per_step'], 'full', d['full_step']['ms_per_step'])
"
