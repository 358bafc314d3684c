#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_switches.py tests/test_gpu_pixellink.py -x -q -m gpu -k "not seed_sweep and not full_size" > gpurun_out/r04_t8.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t8.log
tail -4 gpurun_out/r04_t8.log
python3 scripts/bench_small_conv.py bf16 | tail -17
python3 bench.py --steps 50 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 8 > gpurun_out/r04_b9.json 2> gpurun_out/r04_b9.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b9.json'))
print('f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'full', d['full_step']['ms_per_step'], 'bf16 full', d['bf16']['full_step']['ms_per_step'])
"
