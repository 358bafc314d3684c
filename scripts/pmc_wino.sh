#!/bin/bash
# PMC passes over the Winograd micro-benchmark (one shape): where do the SIMD cycles go?
# usage (on the GPU box): bash scripts/pmc_wino.sh   -> gpurun_out/pmc_wino/*.csv
export SHAPES=${SHAPES:-c:38:512:128}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wino
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python scripts/bench_wino.py > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_wino'
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']:
            k = r['Counter_Name']
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f'{k:36s} {v / max(n, 1):16.0f}  (n={n})')
PY
