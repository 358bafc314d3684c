#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof15 -o fs -- python3 $R/bench.py --dtype bf16 --steps 4 --warmup 2 --steady 0 --cpu-sample 0 --no-input-stage --no-secondary --no-events --full-step 8 > $R/gpurun_out/r04_prof15.log 2>&1
cd $R
python3 scripts/critical_path.py $(ls gpurun_out/r04_prof15/*kernel_trace.csv | head -1) > gpurun_out/r04_cp15.txt 2>&1
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/r04_prof15/fs_kernel_trace.csv'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'pack_input' in r['Kernel_Name']]
step=rows[marks[-2]:marks[-1]]
t0=int(step[0]['Start_Timestamp'])
with open('gpurun_out/r04_step15.txt','w') as f:
    for r in step:
        n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        f.write("%9.1f %8.1f us q%s grid %s,%s,%s %s\n" % ((int(r['Start_Timestamp'])-t0)/1e3, d, r['Queue_Id'], r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'], n[:70]))
PY
