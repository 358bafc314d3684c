#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "mfma:"; python3 scripts/bench_col2im.py 2>/dev/null | head -4
echo "round-3 kernel:"; GSSD_COL2IM_MFMA=0 python3 scripts/bench_col2im.py 2>/dev/null | head -4
python3 scripts/dbg_col2im.py 2>&1 | grep -v "dom new\|dom old"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "col2im or dcn or backward" 2>&1 | tail -3
