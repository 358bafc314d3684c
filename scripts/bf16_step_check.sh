#!/bin/bash
# The bf16 training step on the GPU box: kernel + gradient tests, the step time, and a per-kernel / critical-path profile of one step.
# usage (through gpurun): bash scripts/bf16_step_check.sh [tag]      -> gpurun_out/<tag>_*
tag=${1:-bf16_step}
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "wgrad or backward or training or self_attn or flash or helpers" 2>&1 | tail -8
python3 bench.py --dtype bf16 --steps 10 --warmup 3 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('bf16 fwd + loss ms', d['ms_per_step'], ' full training step ms', d['full_step']['ms_per_step'])
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o fs -- python3 $R/bench.py --dtype bf16 --steps 4 --warmup 2 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
python3 scripts/critical_path.py $(ls gpurun_out/${tag}_prof/*kernel_trace.csv | head -1) "GSSD++ B=32 bf16 storage mode, FULL training step" 5 > gpurun_out/${tag}_critical_path.txt 2>&1
head -5 gpurun_out/${tag}_critical_path.txt
