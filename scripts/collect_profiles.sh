#!/bin/bash
# Round profile collection on the GPU box: bench lines, rocprofv3 kernel stats and the PMC traffic passes -> gpurun_out/rNN_*
# usage: bash scripts/collect_profiles.sh r05   (round 5: bench.py's stdout line is the short one; the full record goes to *_detail.json)
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --detail gpurun_out/${tag}_gssdpp_b32_bench_detail.json > gpurun_out/${tag}_gssdpp_b32_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --detail gpurun_out/${tag}_gssdpp_b32_bf16_bench_detail.json --dtype bf16 --cpu-sample 0 --no-input-stage > gpurun_out/${tag}_gssdpp_b32_bf16_bench.json 2>> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_$dt -o p -- python3 $R/bench.py --full-step 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --steps 20 --warmup 5 --steady 0 --dtype $dt > /dev/null 2>&1
  f=$(find $R/gpurun_out/${tag}_prof_$dt -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/${tag}_gssdpp_b32_${dt}_kernel_stats.csv
done
cd $R
timeout 400 bash scripts/pmc_traffic.sh gssdpp f32 > gpurun_out/${tag}_pmc_f32.log 2>&1
timeout 400 bash scripts/pmc_traffic.sh gssdpp bf16 > gpurun_out/${tag}_pmc_bf16.log 2>&1
timeout 400 bash scripts/pmc_traffic.sh gssd f32 > gpurun_out/${tag}_pmc_gssd.log 2>&1
# layer tables, full training step (bench line + kernel stats), PixelLink++ kernel stats
python3 scripts/layer_times.py gssdpp f32 > gpurun_out/${tag}_gssdpp_b32_f32_layer_times.txt 2>/dev/null
python3 scripts/layer_times.py gssdpp bf16 > gpurun_out/${tag}_gssdpp_b32_bf16_layer_times.txt 2>/dev/null
python3 bench.py --detail gpurun_out/${tag}_gssdpp_b32_fullstep_detail.json --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 8 > gpurun_out/${tag}_gssdpp_b32_fullstep.json 2>> gpurun_out/${tag}_bench.err
python3 bench.py --detail gpurun_out/${tag}_gssd_b32_fullstep_detail.json --config gssd --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --full-step 8 > gpurun_out/${tag}_gssd_b32_fullstep.json 2>> gpurun_out/${tag}_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_fs -o p -- python3 $R/bench.py --steps 2 --warmup 1 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --no-events --full-step 8 > /dev/null 2>&1
f=$(find $R/gpurun_out/${tag}_prof_fs -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_gssdpp_b32_fullstep_kernel_stats.csv
cd $R
# the bf16 training step (bf16 forward, mixed-precision backward): bench line + kernel stats
python3 bench.py --detail gpurun_out/${tag}_gssdpp_b32_bf16_fullstep_detail.json --dtype bf16 --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 8 > gpurun_out/${tag}_gssdpp_b32_bf16_fullstep.json 2>> gpurun_out/${tag}_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_fsb -o p -- python3 $R/bench.py --dtype bf16 --steps 2 --warmup 1 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > /dev/null 2>&1
f=$(find $R/gpurun_out/${tag}_prof_fsb -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_gssdpp_b32_bf16_fullstep_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_pl -o p -- python3 $R/scripts/prof_pixellink.py > $R/gpurun_out/${tag}_pixellink_b32_bench.json 2>/dev/null
f=$(find $R/gpurun_out/${tag}_prof_pl -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_pixellink_b32_kernel_stats.csv
cd $R
tail -3 gpurun_out/${tag}_bench.err
# round 4: per-queue timeline + critical path of one step (fp32, bf16)
cd /tmp
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_cp_$dt -o p -- python3 $R/bench.py --full-step 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --no-events --steps 20 --warmup 5 --steady 0 --dtype $dt > /dev/null 2>&1
  f=$(find $R/gpurun_out/${tag}_cp_$dt -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 $R/scripts/critical_path.py $f "GSSD++ B=32 $dt fwd+loss, hipGraph replay (rocprofv3 --kernel-trace)" 8 > $R/gpurun_out/${tag}_critical_path_$dt.txt
done
cd $R
python3 scripts/host_vs_gpu.py > gpurun_out/${tag}_host_vs_gpu.txt 2>/dev/null
# round 4 (late): the bf16 training step -- critical path of a full step, the GSSD (no attention / DCN) bf16 step, the round-3 form for the A/B
f=$(find $R/gpurun_out/${tag}_prof_fsb -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 scripts/critical_path.py $f "GSSD++ B=32 bf16 storage mode, FULL training step, eager backward (rocprofv3 --kernel-trace)" 5 > gpurun_out/${tag}_critical_path_bf16_fullstep.txt
python3 bench.py --detail gpurun_out/${tag}_gssd_b32_bf16_fullstep_detail.json --config gssd --dtype bf16 --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --full-step 8 > gpurun_out/${tag}_gssd_b32_bf16_fullstep.json 2>> gpurun_out/${tag}_bench.err
GSSD_BWD_BF16=0 python3 bench.py --detail gpurun_out/${tag}_gssdpp_b32_bf16_fullstep_round3_backward_detail.json --dtype bf16 --steps 20 --warmup 5 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > gpurun_out/${tag}_gssdpp_b32_bf16_fullstep_round3_backward.json 2>> gpurun_out/${tag}_bench.err
python3 scripts/bench_wgrad_bf16.py > gpurun_out/${tag}_wgrad_bf16.txt 2>/dev/null
# the driver's own command line
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/${tag}_driver_command_bench_detail.json > gpurun_out/${tag}_driver_command_bench.json 2>/dev/null
# keep the merge small: the raw rocprofv3 output directories stay on the box
rm -rf gpurun_out/${tag}_prof_* gpurun_out/${tag}_cp_*
