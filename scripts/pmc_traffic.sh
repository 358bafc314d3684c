#!/bin/bash
# HBM traffic of every kernel of one bench step (run on the GPU box): two separate --pmc passes (FETCH_SIZE needs 3 of
# the 4 TCC slots, WRITE_SIZE 2), kernel-trace only, as MI355X_MICROARCH.md prescribes.  Writes profiles/pmc_summary.json.
cfg=${1:-gssdpp}
dtype=${2:-f32}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --full-step 0 --steps 3 --warmup 1 --steady 0 --cpu-sample 0 --no-events --no-secondary --no-bf16 --no-input-stage --config $cfg --dtype $dtype > /dev/null 2>&1
done
key=$cfg; [ "$dtype" != f32 ] && key=${cfg}_$dtype
python3 - "$GRAFT_REPO_ROOT" "$key" <<'PY'
import csv, sys, glob, json, collections, re, os
root, cfg = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'{root}/gpurun_out/pmc_{c}/*counter_collection.csv')[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c:
            agg[r['Kernel_Name']][c].append(float(r['Counter_Value']))
def short(n):
    m = re.search(r'conv_igemm_kernel<(\d+), (\d+)', n)
    if m: return f'conv_igemm<{m.group(1)}x{m.group(2)}>'
    m = re.search(r'conv_thin_kernel<(\d+), (\d+)', n)
    if m: return f'conv_thin<{m.group(1)},{m.group(2)}>'
    if 'conv_thin_wino_kernel' in n: return 'conv_thin_wino<16,16>'
    m = re.search(r'conv_wino_x6_kernel<(\d+), (true|false), (\d+)', n)      # <NBT, fused input transform, epilogue (2: pooled), ..>
    if m: return f'conv_wino_x6<{16 * int(m.group(1))}>' + ('' if m.group(2) == 'true' else '/plain') + ('/pool2' if m.group(3) == '2' else '')
    m = re.search(r'flash_attn_x6_kernel<(\d+), (\d+)', n)
    if m: return f'flash_attn_x6<{m.group(1)},{m.group(2)}>'
    m = re.search(r'conv_wino_kernel<(\d+), (true|false), (?:true|false), (\d+)', n)      # <NB, fused input transform, .., epilogue (2: pooled)>
    if m: return f'conv_wino<{m.group(1)}>' + ('' if m.group(2) == 'true' else '/plain') + ('/pool2' if m.group(3) == '2' else '')
    m = re.search(r'conv_bf16_kernel<(\d+), (\d+)', n)
    if m: return f'conv_bf16<{m.group(1)}x{m.group(2)}>'
    m = re.search(r'conv_flat_bf16_kernel<(\d+), (\d+), (?:true|false), (\d+), (\d+)', n)
    if m: return f'conv_flat_bf16<{m.group(1)},{m.group(2)},{64 * int(m.group(4))}>' + ('' if m.group(3) != '2' else '/ring2')
    m = re.search(r'conv_thin_bf16_kernel<(\d+), (\d+)', n)
    if m: return f'conv_thin_bf16<{m.group(1)},{m.group(2)}>'
    m = re.search(r'flash_attn_mixed_kernel<(\d+), (\d+)', n)
    if m: return f'flash_attn_bf16v<{m.group(1)},{m.group(2)}>'
    m = re.search(r'flash_attn_kernel<(\d+), (\d+)', n)
    if m: return f'flash_attn<{m.group(1)},{m.group(2)}>'
    if 'gemm_slot_kernel' in n: return 'gemm_slot<128x128>'
    if 'dcn_fused_kernel' in n: return 'dcn_fused<128x256>'
    if 'dcn_x6_kernel' in n: return 'dcn_x6<128x256>'
    if 'conv_x6_kernel' in n or 'conv_x6_v2_kernel' in n:
        m6 = re.search(r'conv_x6(?:_v2)?_kernel<(\d+)', n)
        return f'conv_x6<{m6.group(1)}>' if m6 else 'conv_x6'
    if 'dcn_bf16_kernel' in n: return 'dcn_bf16<128x256>'
    m = re.search(r'::(\w+_kernel)', n)
    return m.group(1) if m else n[:40]
# template variants of one kernel family share a short name: their launches are pooled (not overwritten)
pool = collections.defaultdict(lambda: collections.defaultdict(list))
for k, d in agg.items():
    for c, v in d.items():
        pool[short(k)][c] += v
out = {}
for k, d in pool.items():
    fs, ws = d.get('FETCH_SIZE', [0]), d.get('WRITE_SIZE', [0])
    f_kb, w_kb = sum(fs) / len(fs), sum(ws) / len(ws)
    # counters are KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM): x2
    out[k] = dict(launches=len(fs), fetch_kib_raw=round(f_kb, 1), write_kib=round(w_kb, 1),
                  hbm_bytes_per_launch=round((2 * f_kb + w_kb) * 1024))
p = os.path.join(root, 'gpurun_out', 'pmc_summary.json')
allj = json.load(open(p)) if os.path.exists(p) else {}
allj[cfg] = out
# the grouped-conv backbone as one figure: every launch of the trunk's conv kernels (thin / flat-window / Winograd families; the
# dilated conv6 runs the 2-stage-ring flat instance and is excluded) plus EVERY BatchNorm + ReLU + pool pass of the step (a slight
# over-count: five of the seventeen passes belong to the fuse convs and extras)
trunk = [k for k in out if k.startswith(('conv_thin', 'conv_wino', 'conv_flat_bf16')) and not k.endswith('/ring2')] + \
        [k for k in out if k.startswith('bn_relu_pool')]
per_step = float(max([out[k]['launches'] for k in out if k.startswith('pack_input')] or [1]))      # the input pack runs once per forward
allj[cfg + '_trunk'] = dict(kernels=sorted(trunk), hbm_bytes_per_step=round(sum(out[k]['hbm_bytes_per_launch'] * out[k]['launches'] for k in trunk) / per_step),
                            note='sum over the trunk kernels of (2 x FETCH_SIZE + WRITE_SIZE) x launches / steps run')
json.dump(allj, open(p, 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
    print(k, v)
PY
