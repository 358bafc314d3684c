#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_pixellink.py tests/test_gpu_multi.py -x -q -m gpu -k "backward or training or autograd or driver or two_ranks or gradient" > gpurun_out/r04_t7.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t7.log
tail -6 gpurun_out/r04_t7.log
python3 bench.py --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 8 > gpurun_out/r04_b4.json 2> gpurun_out/r04_b4.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b4.json'))
print('f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'full', d['full_step']['ms_per_step'], 'host', d['full_step']['host_enqueue_ms_per_step'], 'bf16 full', d['bf16']['full_step']['ms_per_step'], d['bf16']['full_step']['host_enqueue_ms_per_step'])
"
GSSD_NO_BWD_GRAPH=1 python3 bench.py --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 8 > gpurun_out/r04_b5.json 2> gpurun_out/r04_b5.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b5.json'))
print('eager backward: full', d['full_step']['ms_per_step'], 'host', d['full_step']['host_enqueue_ms_per_step'])
"
python3 scripts/thin_knockout.py 0
