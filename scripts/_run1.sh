timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_backward or wgrad_fused or backward_gradients" 2>&1 | tail -3
