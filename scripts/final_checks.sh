cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r06_smoke.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/gpu_suite.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06_driver_command_bench_detail.json > gpurun_out/r06_driver_command_bench.json 2> gpurun_out/r06_bench.err
python3 bench.py --detail gpurun_out/r06_gssdpp_b32_bench_detail.json > gpurun_out/r06_gssdpp_b32_bench.json 2>> gpurun_out/r06_bench.err
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_prof_$dt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --full-step 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --steps 20 --warmup 5 --steady 0 --dtype $dt > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r06_prof_$dt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_gssdpp_b32_${dt}_kernel_stats.csv
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r06_prof_*
tail -3 $GRAFT_REPO_ROOT/gpurun_out/gpu_suite.txt; cat $GRAFT_REPO_ROOT/gpurun_out/r06_smoke.txt | tail -2
