timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -k "graph_replay or end_to_end or self_attn or training_steps or bf16_end" 2>&1 | tail -3
for dt in f32 bf16; do
timeout 400 python bench.py --cpu-sample 0 --no-input-stage --no-secondary --steps 50 --warmup 10 --steady 50 --dtype $dt > gpurun_out/b_$dt.json 2>gpurun_out/b_err.txt
python - <<PY
import json
r=json.loads(open('gpurun_out/b_$dt.json').read().strip().splitlines()[-1])
print('$dt', r['value'], r['ms_per_step'], r.get('steady'))
PY
done
