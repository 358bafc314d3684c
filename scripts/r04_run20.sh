#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -s -k "flash_bwd" 2>&1 | tail -25
