#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1; do
GSSD_TRUNK_PRIORITY=$v python3 bench.py --steps 100 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 0 > gpurun_out/r04_b10_$v.json 2> gpurun_out/r04_b10.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b10_$v.json'))
print('TRUNK_PRIORITY=$v: f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'])
"
done
