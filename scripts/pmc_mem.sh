#!/bin/bash
# memory-path PMC passes over one micro-benchmark; usage (GPU box): bash scripts/pmc_mem.sh <kernel substring> <script.py> [outdir]
pat=$1; script=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/${3:-pmc_mem}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "TD_TC_STALL_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum"; do
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
    print(f'{k:44s} {v / max(n, 1):18.0f}  (n={n})')
PY
