#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over one micro-benchmark script; prints per-counter averages for kernels whose
# name contains $1.   usage (GPU box): bash scripts/pmc_kernel.sh <kernel substring> <script.py> [outdir]
pat=$1; script=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/${3:-pmc_kernel}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $script > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$pat" <<'PY'
import csv, glob, collections, sys
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            k = r['Counter_Name']
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f'{k:36s} {v / max(n, 1):18.0f}  (n={n})')
PY
