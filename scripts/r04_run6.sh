#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1; do
GSSD_BWD_GRAPH=$v python3 bench.py --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 16 > gpurun_out/r04_b6_$v.json 2> gpurun_out/r04_b6.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b6_$v.json'))
print('BWD_GRAPH=$v: full', d['full_step']['ms_per_step'], 'host', d['full_step']['host_enqueue_ms_per_step'])
"
done
