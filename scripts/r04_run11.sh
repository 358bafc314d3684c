#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph_replay or deterministic or end_to_end" 2>&1 | tail -3
for v in 0 1; do
GSSD_BRANCH0_LATE=$v python3 bench.py --steps 100 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --full-step 0 > gpurun_out/r04_b11_$v.json 2> gpurun_out/r04_b11.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b11_$v.json'))
print('BRANCH0_LATE=$v: f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'gssd', d['secondary']['ms_per_step'])
"
done
