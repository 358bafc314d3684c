"""What the memory system gives plain streaming kernels on this MI355X (context for the thin trunk kernels' HBM fractions):
torch's fill (write only), copy (read + write) and sum (read only) on buffers of the conv1_x activation sizes."""
import torch
dev = torch.device('cuda:0')


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (184, 368, 737, 1474):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    tw = timed(lambda: b.fill_(1.0))
    tc = timed(lambda: b.copy_(a))
    tr = timed(lambda: a.sum())
    print(f'{mb:5d} MB: fill {mb / tw / 1e6:5.2f} TB/s written   copy {2 * mb / tc / 1e6:5.2f} TB/s (read + written)   '
          f'sum {mb / tr / 1e6:5.2f} TB/s read')
