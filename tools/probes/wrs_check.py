import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from strajnet_amd.ops import _p, _st, call
F, Hi, Cin, Cout = int(os.environ.get('F', 4)), 128, 96, 48
dt = torch.bfloat16
torch.manual_seed(0)
mode = os.environ.get('MODE', 'rand')
x = torch.randn(F, Hi, Hi, Cin, device='cuda').to(dt)
dp = torch.randn(F, 2 * Hi, 2 * Hi, Cout, device='cuda').to(dt)
if mode == 'rows':      # X = row index, dP = 1 on channel 0: dWeff[.,0,ci] = sum of row indices touched
    x = torch.arange(Hi, device='cuda', dtype=torch.float32).view(1, Hi, 1, 1).expand(F, Hi, Hi, Cin).contiguous().to(dt)
    dp = torch.zeros_like(dp); dp[..., 0] = 1
if mode == 'cols':
    x = torch.arange(Hi, device='cuda', dtype=torch.float32).view(1, 1, Hi, 1).expand(F, Hi, Hi, Cin).contiguous().to(dt)
    dp = torch.zeros_like(dp); dp[..., 0] = 1
if mode == 'ones':
    x = torch.ones_like(x); dp = torch.zeros_like(dp); dp[..., 0] = 1
dweff = torch.zeros(16, Cout, Cin, device='cuda')
dbp = torch.zeros(32, Cout, device='cuda')
call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(dbp), 32, F, Hi, Hi, Cin, Cout, 256, 1, _st())
torch.cuda.synchronize()
xd, dd = x.double(), dp.double()
xp = torch.nn.functional.pad(xd, (0, 0, 1, 1, 1, 1))       # pad cols and rows by 1
ref = torch.zeros(16, Cout, Cin, dtype=torch.float64, device='cuda')
for a in range(2):
    for b in range(2):
        d = dd[:, a::2, b::2, :]                         # [F,Hi,Wi,Cout]
        for r in range(2):
            for s in range(2):
                xs = xp[:, a + r:a + r + Hi, b + s:b + s + Hi, :]      # X[i+a-1+r, j+b-1+s]
                ref[a * 8 + b * 4 + r * 2 + s] = torch.einsum('fijo,fijc->oc', d, xs)
err = (dweff.double() - ref)
for pt in range(16):
    print(pt, 'a,b,r,s=', pt >> 3, (pt >> 2) & 1, (pt >> 1) & 1, pt & 1, 'max|err| %.4g' % err[pt].abs().max().item(), ' ref max %.4g' % ref[pt].abs().max().item(),
          ' got[0,0] %.6g ref[0,0] %.6g' % (dweff[pt, 0, 0].item(), ref[pt, 0, 0].item()))
dbr = dd.sum((0, 1, 2))
print('db err', (dbp.double().sum(0) - dbr).abs().max().item(), dbr.abs().max().item())
print('db got', dbp.double().sum(0)[:8].tolist()); print('db ref', dbr[:8].tolist())
for a in range(2):
    for b in range(2): print('ref part a,b', a, b, dd[:, a::2, b::2, :].sum((0,1,2))[:4].tolist())
for name, mk in (('chan', lambda d: d.copy_(torch.arange(1, Cout + 1, device='cuda', dtype=torch.float32).view(1, 1, 1, Cout).expand_as(d))),
                 ('par10', lambda d: (d.zero_(), d[:, 1::2, 0::2, :].fill_(1))),
                 ('rowid', lambda d: d.copy_(torch.arange(2 * Hi, device='cuda', dtype=torch.float32).view(1, 2 * Hi, 1, 1).expand_as(d) % 7)),
                 ('colid', lambda d: d.copy_(torch.arange(2 * Hi, device='cuda', dtype=torch.float32).view(1, 1, 2 * Hi, 1).expand_as(d) % 5))):
    mk(dp)
    dweff.zero_(); dbp.zero_()
    call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(dbp), 32, F, Hi, Hi, Cin, Cout, 256, 1, _st())
    torch.cuda.synchronize()
    print(name, 'got', dbp.double().sum(0)[:6].tolist(), 'ref', dp.double().sum((0, 1, 2))[:6].tolist())
bad = []
for cs in list(range(0, 70)) + [126, 127, 128, 129, 254, 255]:
    dp.zero_(); dp[:, :, cs, :] = 1
    dweff.zero_(); dbp.zero_()
    call('stj_upconv_wgrad', _p(x), _p(dp), _p(dweff), _p(dbp), 32, F, Hi, Hi, Cin, Cout, 256, 1, _st())
    torch.cuda.synchronize()
    v = dbp.double().sum(0)[0].item()
    if v != 4 * 256: bad.append((cs, v))
print('single-column db: wrong at', bad)
