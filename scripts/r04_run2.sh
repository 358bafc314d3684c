#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "streamk or full_size_properties or graph_replay or deterministic" > gpurun_out/r04_t4.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t4.log
tail -4 gpurun_out/r04_t4.log
python3 bench.py --steps 50 --warmup 10 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 0 > gpurun_out/r04_b1.json 2> gpurun_out/r04_b1.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b1.json'))
print('f32 ms', d['ms_per_step'], 'bf16 ms', d['bf16']['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'bf16 trunk', d['bf16']['roofline']['frac'], d['bf16']['roofline']['ms_per_step'])
"
./scripts/ubench/thin_skeleton > gpurun_out/r04_thin_skeleton.txt 2>&1
cat gpurun_out/r04_thin_skeleton.txt
