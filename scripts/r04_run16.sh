#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "conv_wgrad_bf16" 2>&1 | tail -15
timeout 300 python3 scripts/bench_wgrad_bf16.py 2>&1 | tail -12
