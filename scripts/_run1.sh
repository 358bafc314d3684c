timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_backward or wgrad_fused or backward_gradients or vanilla" 2>&1 | tail -3
timeout 400 python bench.py --cpu-sample 0 --no-input-stage --no-secondary --no-events --steps 2 --warmup 1 --steady 0 --full-step 8 > gpurun_out/fs_bench.json 2>gpurun_out/fs_err.txt
python - <<'PY'
import json
print(json.loads(open('gpurun_out/fs_bench.json').read().strip().splitlines()[-1])['full_step'])
PY
