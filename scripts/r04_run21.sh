#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "bn_backward_mixed or helpers_of_the_backward" 2>&1 | tail -15
