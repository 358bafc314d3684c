#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "wgrad or backward or training or self_attn" 2>&1 | tail -8
GSSD_BWD_BF16=1 python3 bench.py --dtype bf16 --steps 10 --warmup 3 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > gpurun_out/r04_b15_1.json 2> gpurun_out/r04_b15.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b15_1.json'))
print('bf16 fwd ms', d['ms_per_step'], 'full', d['full_step']['ms_per_step'])
"
bash scripts/r04_run18.sh
head -5 gpurun_out/r04_cp15.txt
