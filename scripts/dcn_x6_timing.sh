#!/bin/bash
# Per-phase wall-clock breakdown of dcn_x6's K loop (debug build -DX6_TIMING through GSSD_LIB_PATH; wave 0 of every workgroup accumulates the
# 100-MHz real-time ticks between its phase boundaries).  usage (GPU box): bash scripts/dcn_x6_timing.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/grouped-ssd-pytorch_amd/gssd/csrc
OBJS=$(ls *.o | grep -v '^dcn_x6.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -I. -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize -DX6_TIMING -DX6_TWAVE=${TWAVE:-0} $EXTRA -c dcn_x6.hip -o /tmp/x6_t.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/x6_t.o -o /tmp/libgssd_x6_t.so &&
GSSD_LIB_PATH=/tmp/libgssd_x6_t.so python3 - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, H, Cc, dg, Cout = 32, 38, 1024, 4, 512
x = torch.randn(B, H, H, Cc, device=dev)
om = torch.randn(B, H, H, 27 * dg, device=dev) * 0.5
w = torch.randn(Cout, Cc, 3, 3, device=dev) * 0.02
bias = torch.randn(Cout, device=dev)
rd = C.CDLL(_lib.LIB_PATH).gssd_dcn_x6_timing_read
buf = (C.c_ulonglong * 8)()
for _ in range(2): ops.dcn_forward_x6(x, om, w, bias, dg)
torch.cuda.synchronize(); rd(buf)
n = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): ops.dcn_forward_x6(x, om, w, bias, dg)
e1.record(); torch.cuda.synchronize(); rd(buf)
t = [v / n for v in buf]
wgs, its = 722, 288
tw = int(os.environ.get('TWAVE', '0'))
if tw < 8:
    names = ['part A + first group of part B', 'M: wait lgkmcnt + barrier', 'part B groups 1..7', 'E: barrier', '-', '-', '-', '-']
else:
    names = ['DMA issue, wait corners, blend + split', 'M: wait Y pieces + barrier', 'plane writes, corner requests', 'E: wait X pieces, LDS writes + barrier', '-', '-', '-', '-']
print(f'dcn_x6 (timing build): {e0.elapsed_time(e1) / n:.3f} ms per launch incl. the weight split; wave {tw} of {wgs} workgroups, {its} iterations each')
tot = sum(t)
for k in range(8):
    per = t[k] / wgs * 10.0          # ns per workgroup
    if t[k]: print(f'  {names[k]:48s} {100 * t[k] / tot:5.1f} %   {per / its:9.1f} ns per iteration')
PY
