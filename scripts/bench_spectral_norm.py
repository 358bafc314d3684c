"""Time of the spectral-norm launch of GSSD++ (one workgroup per matrix: 8 Self_Attn blocks x theta | phi | g | attn projections)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
items = []
for C in (512, 1024, 512, 256, 256, 256, 512, 1024):          # channels of the attention blocks (38 x 38 x 2, 19 x 19, 10 x 10, ... + SA-base)
    for (r, c) in ((C // 8, C), (C // 8, C), (C // 2, C), (C, C // 2)):
        w = torch.randn(r, c, 1, 1, device=dev) * 0.05
        items.append((w, torch.nn.functional.normalize(torch.randn(r, device=dev), dim=0), torch.nn.functional.normalize(torch.randn(c, device=dev), dim=0),
                      torch.zeros(r, device=dev)))
tab = ops.sn_items_tensor(items, dev)
for it in (True, False):
    for _ in range(3):
        ops.spectral_norm(tab, len(items), it)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.spectral_norm(tab, len(items), it)
    e1.record()
    torch.cuda.synchronize()
    print(f'spectral norm, {len(items)} matrices, power iteration {it}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us')
