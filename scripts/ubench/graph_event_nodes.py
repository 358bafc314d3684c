import torch
dev=torch.device('cuda:0')
a=torch.randn(4096,4096,device=dev); b=torch.randn(4096,4096,device=dev)
e0=torch.cuda.Event(enable_timing=True, external=True); e1=torch.cuda.Event(enable_timing=True, external=True)
s=torch.cuda.Stream()
torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    c=a@b
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        d=a+b
        e0.record()
        c=a@b
        e1.record()
        f=c+1
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('elapsed in graph', e0.elapsed_time(e1), 'ms')
t0=torch.cuda.Event(enable_timing=True); t1=torch.cuda.Event(enable_timing=True)
t0.record(); c=a@b; t1.record(); torch.cuda.synchronize(); print('eager', t0.elapsed_time(t1))
