#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 scripts/bench_col2im.py 2>/dev/null | head -4
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "col2im or dcn or backward" 2>&1 | tail -5
