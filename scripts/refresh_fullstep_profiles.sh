#!/bin/bash
# Re-collects the bench lines and the bf16 training-step files of a profile set after a change that only touches the backward.
# usage (through gpurun): bash scripts/refresh_fullstep_profiles.sh r04_d
tag=${1:-r04_d}
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > gpurun_out/${tag}_gssdpp_b32_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --dtype bf16 --cpu-sample 0 --no-input-stage > gpurun_out/${tag}_gssdpp_b32_bf16_bench.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --dtype bf16 --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 8 > gpurun_out/${tag}_gssdpp_b32_bf16_fullstep.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --config gssd --dtype bf16 --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 8 > gpurun_out/${tag}_gssd_b32_bf16_fullstep.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 8 > gpurun_out/${tag}_gssdpp_b32_fullstep.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --config gssd --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 8 > gpurun_out/${tag}_gssd_b32_fullstep.json 2>> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof_fsb $R/gpurun_out/${tag}_prof_fs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_fsb -o p -- python3 $R/bench.py --dtype bf16 --steps 2 --warmup 1 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > /dev/null 2>&1
f=$(find $R/gpurun_out/${tag}_prof_fsb -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_gssdpp_b32_bf16_fullstep_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_fs -o p -- python3 $R/bench.py --steps 2 --warmup 1 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --no-events --full-step 8 > /dev/null 2>&1
f=$(find $R/gpurun_out/${tag}_prof_fs -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_gssdpp_b32_fullstep_kernel_stats.csv
cd $R
f=$(find gpurun_out/${tag}_prof_fsb -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 scripts/critical_path.py $f "GSSD++ B=32 bf16 storage mode, FULL training step, eager backward (rocprofv3 --kernel-trace)" 5 > gpurun_out/${tag}_critical_path_bf16_fullstep.txt
f=$(find gpurun_out/${tag}_prof_fs -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 scripts/critical_path.py $f "GSSD++ B=32 fp32, FULL training step, eager backward (rocprofv3 --kernel-trace)" 5 > gpurun_out/${tag}_critical_path_f32_fullstep.txt
tail -2 gpurun_out/${tag}_bench.err
