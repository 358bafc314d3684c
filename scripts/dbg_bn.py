import torch, torch.nn.functional as F
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
for shape in ((4, 64, 300, 300), (4, 512, 38, 38), (4, 256, 1, 1), (4, 1024, 19, 19)):
    x = torch.randn(*shape) * 2 + 0.5; w = torch.rand(shape[1]) + 0.5; b = torch.randn(shape[1])
    ref = F.batch_norm(x.double(), None, None, w.double(), b.double(), True, 0.0, 1e-5)
    y1 = F.batch_norm(x.cuda(), None, None, w.cuda(), b.cuda(), True, 0.0, 1e-5)
    rm, rv = torch.zeros(shape[1]).cuda(), torch.ones(shape[1]).cuda()
    y2 = F.batch_norm(x.cuda(), rm, rv, w.cuda(), b.cuda(), True, 0.1, 1e-5)
    torch.backends.cudnn.enabled = False
    y3 = F.batch_norm(x.cuda(), None, None, w.cuda(), b.cuda(), True, 0.0, 1e-5)
    torch.backends.cudnn.enabled = True
    yc = F.batch_norm(x, None, None, w, b, True, 0.0, 1e-5)
    print(shape, 'gpu none-stats', f'{rel(y1, ref):.1e}', 'gpu with stats', f'{rel(y2, ref):.1e}', 'gpu cudnn off', f'{rel(y3, ref):.1e}', 'cpu', f'{rel(yc, ref):.1e}')
x = torch.randn(4, 64, 75, 75)
print('maxpool ceil', rel(F.max_pool2d(x.cuda(), 2, 2, 0, ceil_mode=True), F.max_pool2d(x, 2, 2, 0, ceil_mode=True)))
