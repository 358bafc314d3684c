import torch, torch.nn.functional as F, time
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
for (cin, cout, g, H, k) in ((64, 64, 4, 150, 3), (512, 512, 4, 38, 3), (512, 512, 1, 38, 1), (12, 64, 4, 300, 3)):
    x = torch.randn(4, cin, H, H); w = torch.randn(cout, cin // g, k, k) * 0.05
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2, 1, g)
    for name, setup in (('default', lambda: None), ('cudnn_off', lambda: setattr(torch.backends.cudnn, 'enabled', False)),
                        ('cudnn_on_det', lambda: (setattr(torch.backends.cudnn, 'enabled', True), setattr(torch.backends.cudnn, 'deterministic', True))),
                        ('channels_last', lambda: setattr(torch.backends.cudnn, 'deterministic', False))):
        setup()
        xd, wd = x.cuda(), w.cuda()
        if name == 'channels_last':
            xd = xd.contiguous(memory_format=torch.channels_last); wd = wd.contiguous(memory_format=torch.channels_last)
        y = F.conv2d(xd, wd, None, 1, k // 2, 1, g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): y = F.conv2d(xd, wd, None, 1, k // 2, 1, g)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print((cin, cout, g, H, k), name, 'rel err vs fp64', f'{rel(y, ref):.2e}', f'{dt*1e6:.0f} us')
    print('  cpu fp32 err', f'{rel(F.conv2d(x, w, None, 1, k // 2, 1, g), ref):.2e}')
