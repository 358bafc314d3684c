"""Does a timing event recorded INSIDE a captured hipGraph work on this ROCm / torch (torch.cuda.Event(external=True))?"""
import torch
dev = torch.device('cuda:0')
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty_like(a)
for _ in range(2): torch.mm(a, b, out=c)
torch.cuda.synchronize()
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print('external kwarg unsupported', ex); raise SystemExit
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        c.add_(1.0)
        e0.record()
        torch.mm(a, b, out=c)
        e1.record()
        c.mul_(0.5)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        print('elapsed inside graph ms', e0.elapsed_time(e1))
except Exception as ex:
    print('FAILED', type(ex).__name__, ex)
s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
s.record(); torch.mm(a, b, out=c); t.record(); torch.cuda.synchronize(); print('eager mm ms', s.elapsed_time(t))
