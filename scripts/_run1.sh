R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fs_prof -o p -- python3 $R/bench.py --cpu-sample 0 --no-input-stage --no-secondary --no-events --steps 2 --warmup 1 --steady 0 --full-step 6 > $R/gpurun_out/fs_bench.json 2>$R/gpurun_out/fs_err.txt
f=$(find $R/gpurun_out/fs_prof -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/fs_kernel_stats.csv
