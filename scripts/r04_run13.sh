#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamk or dcn_fused" 2>&1 | tail -3
for b in 32 24 12; do B=$b python3 scripts/bench_dcn.py 2>/dev/null | tail -1; done
GSSD_DCN_STREAMK=0 python3 scripts/bench_dcn.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r04_pmc_dcn_$c -o p -- python3 $R/scripts/bench_dcn.py > /dev/null 2>&1
python3 - "$R/gpurun_out/r04_pmc_dcn_$c" $c <<'PY'
import csv, glob, sys
v=[float(r['Counter_Value']) for f in glob.glob(sys.argv[1]+'/*counter_collection.csv') for r in csv.DictReader(open(f)) if 'dcn_fused' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[2]]
print(sys.argv[2], 'per launch (raw counter, KB):', sum(v)/len(v), 'n', len(v))
PY
done
