timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "col2im or backward_gradients or test_conv_backward or sa_backward" 2>&1 | tail -4
timeout 400 python bench.py --cpu-sample 0 --no-input-stage --no-secondary --no-events --steps 2 --warmup 1 --steady 0 --full-step 8 > gpurun_out/fs_bench.json 2>gpurun_out/fs_err.txt
python - <<'PY'
import json
print(json.loads(open('gpurun_out/fs_bench.json').read().strip().splitlines()[-1])['full_step'])
PY
