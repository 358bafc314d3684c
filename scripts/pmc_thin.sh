#!/bin/bash
# SQ / TCC / TCP counter passes over the bf16 bench for the thin trunk kernels (run on the GPU box): kernel-trace + --pmc only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_thin_$i -o p -- python3 $R/bench.py --full-step 0 --steps 3 --warmup 1 --steady 0 --cpu-sample 0 --no-events --no-secondary --no-input-stage --dtype bf16 > /dev/null 2>&1
  echo "pass $i rc $?"
done
python3 - "$R" <<'PY'
import csv, sys, glob, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f'{root}/gpurun_out/pmc_thin_*')):
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'conv_thin_bf16' in n or 'conv_bf16_kernel<128, 64' in n or 'conv_flat_bf16' in n or 'dcn_bf16' in n or 'flash_attn_mixed' in n:
                k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in cs.items():
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
