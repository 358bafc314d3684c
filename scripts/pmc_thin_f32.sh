#!/bin/bash
# SQ counter passes over the fp32 bench for the trunk kernels (run on the GPU box): kernel-trace + --pmc only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_thin32_$i -o p -- python3 $R/bench.py --full-step 0 --steps 3 --warmup 1 --steady 0 --cpu-sample 0 --no-events --no-secondary --no-input-stage --no-bf16 > /dev/null 2>&1
  echo "pass $i rc $?"
done
python3 - "$R" <<'PY'
import csv, sys, glob, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f'{root}/gpurun_out/pmc_thin32_*')):
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'conv_thin' in n or 'conv_wino' in n or 'dcn_fused' in n or 'flash_attn' in n or 'gemm_slot' in n:
                k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    w = sum(cs.get('SQ_WAVE_CYCLES', [0])) / max(len(cs.get('SQ_WAVE_CYCLES', [1])), 1)
    for c, v in cs.items():
        m = sum(v) / len(v)
        print(f'   {c:30s} n={len(v):3d} mean={m:.4g}' + (f'   {100 * m / w:5.1f} % of wave cycles' if w and c.startswith(('SQ_ACTIVE', 'SQ_WAIT')) else ''))
PY
