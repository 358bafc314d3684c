import os, sys, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'grouped-ssd-pytorch_amd'))
import torch
if len(sys.argv) > 1:
    from gssd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    B, H, C, dg = 1, 11, 128, 2
    x = torch.randn(B, H, H, C, device=dev)
    om = torch.randn(B, H, H, 27 * dg, device=dev) * float(sys.argv[2])
    dcols = torch.randn(B * H * H, 9 * C, device=dev)
    dx = torch.zeros_like(x); dom = torch.zeros_like(om)
    ops.dcn_col2im(x, om, dcols, dx, dom, dg)
    torch.save((dx.cpu(), dom.cpu()), sys.argv[1])
else:
    for std in ('0.0', '0.4', '2.5'):
        for v, f in (('1', '/tmp/c_new.pt'), ('0', '/tmp/c_old.pt')):
            subprocess.run([sys.executable, __file__, f, std], env=dict(os.environ, GSSD_COL2IM_MFMA=v), check=True, stderr=subprocess.DEVNULL)
        a, b = torch.load('/tmp/c_new.pt'), torch.load('/tmp/c_old.pt')
        ddx, ddom = (a[0] - b[0]).abs(), (a[1] - b[1]).abs()
        print('std', std, 'dx maxdiff', float(ddx.max()), 'of', float(b[0].abs().max()), '| dom maxdiff', float(ddom.max()), 'of', float(b[1].abs().max()))
        print('  dom new', a[1][0, 3, 3, :10].tolist()); print('  dom old', b[1][0, 3, 3, :10].tolist()); print('  dom new nonzero', int((a[1] != 0).sum()), 'old', int((b[1] != 0).sum()))
        bad = (ddx > 1e-3).nonzero()
        print('  bad dx entries', bad.shape[0], 'of', ddx.numel())
        if bad.shape[0]:
            print('  pixels (y,x) with errors:', sorted({(int(i[1]), int(i[2])) for i in bad})[:40])
            print('  channels with errors:', sorted({int(i[3]) for i in bad})[:70])
