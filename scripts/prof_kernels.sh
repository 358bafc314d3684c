#!/bin/bash
# rocprofv3 kernel stats of one bench.py run on the GPU box: top kernels by total time.
# usage: bash scripts/prof_kernels.sh <tag> [bench.py flags...]   -> gpurun_out/<tag>_kernel_stats.csv (+ the table on stdout)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o p -- python3 $R/bench.py --full-step 0 --cpu-sample 0 --no-input-stage --no-secondary --no-bf16 --steps 20 --warmup 5 --steady 0 "$@" > $R/gpurun_out/${tag}_prof_bench.json 2> /dev/null
f=$(find $R/gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats.csv
rm -rf $R/gpurun_out/${tag}_prof
cd $R
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
for r in rows[:${TOP:-18}]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
try:
    print('ms_per_step', json.loads(open("gpurun_out/${tag}_prof_bench.json").read().strip().splitlines()[-1])['ms_per_step'])
except Exception as e:
    print('bench line:', e)
PY
