import torch, torch.nn.functional as F
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
cases = [('conv6', 512, 1024, 4, 19, 3, 1, 6, 6), ('conv7', 1024, 1024, 4, 19, 1, 1, 0, 1), ('ext1', 256, 512, 4, 19, 3, 2, 1, 1),
         ('ext0', 1024, 256, 4, 19, 1, 1, 0, 1), ('ext3', 128, 256, 4, 10, 3, 2, 1, 1), ('ext5', 128, 256, 4, 5, 3, 1, 0, 1),
         ('head1', 1024, 36, 1, 19, 3, 1, 1, 1), ('fuse21', 1024, 1024, 1, 19, 1, 1, 0, 1), ('conv1_1', 12, 64, 4, 300, 3, 1, 1, 1),
         ('conv5', 512, 512, 4, 19, 3, 1, 1, 1)]
for (n, cin, cout, g, H, k, s, p, d) in cases:
    x = torch.randn(4, cin, H, H); w = torch.randn(cout, cin // g, k, k) * 0.05; b = torch.randn(cout)
    ref = F.conv2d(x.double(), w.double(), b.double(), s, p, d, g)
    y = F.conv2d(x.cuda(), w.cuda(), b.cuda(), s, p, d, g)
    print(n, f'{rel(y, ref):.2e}')
