#!/bin/bash
# Same-box A/B of round 4's dcn_bf16 kernel (the default) and round 5's loader + matrix-wave form (GSSD_DCN_BF16_V3=1), with a bit-for-bit comparison of
# the outputs (same blend arithmetic, same K order).  usage (GPU box): bash scripts/dcn_bf16_ab.sh
cd "$(dirname "$0")/.."
DUMP=/tmp/dcnb_old python scripts/bench_dcn_bf16.py 2>&1 | grep "dcn_bf16 B" | sed 's/^/round 4: /'
GSSD_DCN_BF16_V3=1 DUMP=/tmp/dcnb_new python scripts/bench_dcn_bf16.py 2>&1 | grep "dcn_bf16 B" | sed 's/^/round 5: /'
python - <<'PY'
import torch
for B in (11, 22, 32):
    a, b = torch.load(f'/tmp/dcnb_old_{B}.pt'), torch.load(f'/tmp/dcnb_new_{B}.pt')
    print(f'B={B}:', 'outputs bit-identical' if torch.equal(a, b) else f'max abs diff {float((a.float() - b.float()).abs().max()):.3e}')
PY
