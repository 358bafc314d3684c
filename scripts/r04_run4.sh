#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_sum_replicas or end_to_end or pooled_raw" > gpurun_out/r04_t5.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t5.log
tail -5 gpurun_out/r04_t5.log
python3 scripts/layer_times.py gssdpp bf16 2>/dev/null | head -16
python3 scripts/layer_times.py gssdpp f32 2>/dev/null | head -16
python3 bench.py --steps 50 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 4 > gpurun_out/r04_b3.json 2> gpurun_out/r04_b3.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b3.json'))
print('f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'bf16 trunk', d['bf16']['roofline']['frac'], d['bf16']['roofline']['ms_per_step'], 'full', d['full_step']['ms_per_step'], 'bf16 full', d['bf16']['full_step'])
"
python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_t6.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t6.log
tail -5 gpurun_out/r04_t6.log
