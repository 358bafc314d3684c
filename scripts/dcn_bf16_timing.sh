#!/bin/bash
# Per-phase wall-clock breakdown of dcn_bf16's round-5 kernel (debug build -DX6_TIMING through GSSD_LIB_PATH; wave TWAVE of every workgroup
# accumulates the 100-MHz real-time ticks between its phase boundaries).  usage (GPU box): TWAVE=8 bash scripts/dcn_bf16_timing.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/grouped-ssd-pytorch_amd/gssd/csrc
OBJS=$(ls *.o | grep -v '^dcn_bf16.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -I. -munsafe-fp-atomics -Wno-unused-result -DX6_TIMING -DX6_TWAVE=${TWAVE:-0} $EXTRA -c dcn_bf16.hip -o /tmp/dcnb_t.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dcnb_t.o -o /tmp/libgssd_dcnb_t.so &&
GSSD_DCN_BF16_V3=1 GSSD_LIB_PATH=/tmp/libgssd_dcnb_t.so python3 - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'grouped-ssd-pytorch_amd'))
import torch
from gssd import _lib
from gssd._lib import lib, check
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, H, Cc, dg, Cout = int(os.environ.get('B', 32)), 38, 1024, 4, 512
w = torch.randn(Cout, Cc, 3, 3, device=dev) * 0.01
bias = torch.randn(Cout, device=dev)
wp = torch.empty(int(lib.gssd_dcn_packed_weight_elems_bf16(Cout, Cc)), device=dev, dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
check(lib.gssd_dcn_pack_weight_bf16(w.data_ptr(), wp.data_ptr(), Cout, Cc, dg, s))
x = torch.randn(B, H, H, Cc, device=dev).to(torch.bfloat16)
om = torch.randn(B, H, H, 27 * dg, device=dev) * 0.8
out = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
def run(): check(lib.gssd_dcn_forward_bf16(x.data_ptr(), om.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, H, Cc, dg, 27 * dg, Cout, s))
rd = C.CDLL(_lib.LIB_PATH).gssd_dcn_bf16_timing_read
buf = (C.c_ulonglong * 8)()
for _ in range(2): run()
torch.cuda.synchronize(); rd(buf)
n = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize(); rd(buf)
t = [v / n for v in buf]
wgs, its = ((B * H * H + 127) // 128) * 2, 288
tw = int(os.environ.get('TWAVE', '0'))
names = (['part A + first group of part B', 'M: wait lgkmcnt + barrier', 'part B groups 1..7', 'E: barrier'] if tw < 8 else
         ['DMA issue, wait corners, blend', 'M: wait Y pieces + barrier', 'plane writes, corner requests', 'E: wait X pieces, LDS writes + barrier'])
print(f'dcn_bf16 (timing build): {e0.elapsed_time(e1) / n * 1e3:.0f} us per launch; wave {tw} of {wgs} workgroups, {its} iterations each')
tot = sum(t)
for k in range(4):
    print(f'  {names[k]:48s} {100 * t[k] / tot:5.1f} %   {t[k] / wgs * 10.0 / its:9.1f} ns per iteration')
PY
