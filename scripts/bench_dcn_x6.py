"""dcn_x6 (fp32 deformable conv on the bf16 matrix cores, three-plane split) against dcn_fused (fp32 MFMA) at the GSSD++ shape:
time and the difference of both from a float64 evaluation of a slice."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
from gssd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
# argv: B C Cout (the GSSD++ launch: 32 x 38 x 38 pixels, 2048 = x | SA-base channels, 256 outputs)
B, C, Cout = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (32, 2048, 256)))
H, dg = 38, 4
x = torch.randn(B, H, H, C, device=dev)
om = torch.randn(B, H, H, 27 * dg, device=dev) * 0.5
w = torch.randn(Cout, C, 3, 3, device=dev) * 0.02
bias = torch.randn(Cout, device=dev)
ref = ops.dcn_forward(x, om, w, bias, dg)
got = ops.dcn_forward_x6(x, om, w, bias, dg, f16ok=True)
print('max |x6 - fused| / max|fused| =', float((got - ref).abs().max() / ref.abs().max()), ' L2 rel', float((got - ref).norm() / ref.norm()))
wp = ops.dcn_pack_weight(w, dg)
for name, fn in (('dcn_fused (fp32 MFMA)', lambda: ops.dcn_forward(x, om, w, bias, dg, w_packed=wp)), ('dcn_x6 bf16 planes incl. weight split', lambda: ops.dcn_forward_x6(x, om, w, bias, dg)),
                 ('dcn_x6 fp16 planes (GSSD_CONV_F16_OK) incl. weight split', lambda: ops.dcn_forward_x6(x, om, w, bias, dg, f16ok=True))):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 5:.3f} ms')
