"""dcn_bf16 on the GSSD++ shape (38x38, 1024 -> 512, 4 deformable groups) at several batch sizes: 11 (250 tiles: every workgroup alone on
its CU), 22 (498 tiles: one round of co-resident pairs), 32 (722 tiles: the bench shape)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops
from gssd._lib import lib, check
dev = torch.device('cuda:0')
H, C, Cout, dg = 38, 1024, 512, 4
torch.manual_seed(0)
w = torch.randn(Cout, C, 3, 3, device=dev) * 0.01
bias = torch.randn(Cout, device=dev)
n_el = int(lib.gssd_dcn_packed_weight_elems_bf16(Cout, C))
wp = torch.empty(n_el, device=dev, dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
check(lib.gssd_dcn_pack_weight_bf16(w.data_ptr(), wp.data_ptr(), Cout, C, dg, s))
for B in (11, 22, 32):
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    om = torch.randn(B, H, H, 27 * dg, device=dev) * 0.8
    out = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
    def run():
        check(lib.gssd_dcn_forward_bf16(x.data_ptr(), om.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, H, C, dg, 27 * dg, Cout, s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tiles = ((B * H * H + 127) // 128) * 2
    fl = 2.0 * B * H * H * Cout * 9 * C
    print(f'dcn_bf16 B={B}: {tiles} tiles, {ms * 1e3:.0f} us, {fl / ms / 1e9:.0f} TFLOP/s, checksum {float(out.float().double().sum()):.6f} {float(out.float().double().abs().sum()):.6f}')
    if os.environ.get('DUMP'): torch.save(out.cpu(), os.environ['DUMP'] + f'_{B}.pt')
