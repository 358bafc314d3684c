#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 scripts/layer_times.py gssdpp bf16 2>/dev/null | head -14
python3 scripts/layer_times.py gssdpp f32 2>/dev/null | head -6
python3 bench.py --steps 50 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 0 > gpurun_out/r04_b2.json 2> gpurun_out/r04_b2.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b2.json'))
print('f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'bf16 trunk', d['bf16']['roofline']['frac'], d['bf16']['roofline']['ms_per_step'])
"
