#!/bin/bash
# round 4, call 1: new parity / stream-K / 8-rank tests, critical-path traces (fp32 + bf16), then the whole -m gpu suite
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "full_size_parity or streamk" > gpurun_out/r04_t1.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t1.log
python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu -s > gpurun_out/r04_t2.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t2.log
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_cp_$dt -o p -- python3 $R/bench.py --full-step 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --no-events --steps 20 --warmup 5 --steady 0 --dtype $dt > /dev/null 2>&1
  f=$(find $R/gpurun_out/r04_cp_$dt -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 $R/scripts/critical_path.py $f "GSSD++ B=32 $dt fwd+loss, hipGraph replay (rocprofv3 --kernel-trace)" 8 > $R/gpurun_out/r04_critical_path_$dt.txt
done
cd $R
python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_t3.log 2>&1
echo "rc=$?" >> gpurun_out/r04_t3.log
tail -5 gpurun_out/r04_t1.log gpurun_out/r04_t2.log gpurun_out/r04_t3.log
