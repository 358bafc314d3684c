#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -s -k "wgrad_nt or backward or training" 2>&1 | tail -25
for v in 0 1; do
GSSD_BWD_BF16=$v python3 bench.py --dtype bf16 --steps 10 --warmup 3 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > gpurun_out/r04_b15_$v.json 2> gpurun_out/r04_b15.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04_b15_$v.json'))
print('BWD_BF16=$v: bf16 fwd ms', d['ms_per_step'], 'full', d['full_step']['ms_per_step'])
"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof15 -o fs -- python3 $R/bench.py --dtype bf16 --steps 4 --warmup 2 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > $R/gpurun_out/r04_prof15.log 2>&1
cd $R
ls gpurun_out/r04_prof15 | head
python3 scripts/critical_path.py $(ls gpurun_out/r04_prof15/*kernel_trace.csv | head -1) > gpurun_out/r04_cp15.txt 2>&1
tail -5 gpurun_out/r04_cp15.txt
