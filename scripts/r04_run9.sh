#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04b_cp_$dt -o p -- python3 $R/bench.py --full-step 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --no-events --steps 20 --warmup 5 --steady 0 --dtype $dt > /dev/null 2>&1
  f=$(find $R/gpurun_out/r04b_cp_$dt -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 $R/scripts/critical_path.py $f "GSSD++ B=32 $dt fwd+loss, hipGraph replay (rocprofv3 --kernel-trace)" 8 > $R/gpurun_out/r04b_critical_path_$dt.txt
done
head -12 $R/gpurun_out/r04b_critical_path_f32.txt
