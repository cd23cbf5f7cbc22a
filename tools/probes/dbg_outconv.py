import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
sys.path.insert(0, 'tests')
from test_ops_gpu import mk_param, rnd, ref_of
from strajnet_amd import ops
dt = torch.bfloat16
B, Tn, H, C = 2, 8, 32, 48
ps = [mk_param((3, 3, C, 2), dt, 0.1, 1), mk_param((2,), dt, 0.1, 2), mk_param((3, 3, C, 2), dt, 0.1, 3), mk_param((2,), dt, 0.1, 4)]
xo = rnd((B * Tn, H, H, C), dt, 5).requires_grad_(True)
xf = rnd((B * Tn, H, H, C), dt, 6).requires_grad_(True)
out = ops.outconv_pair(xo, xf, *ps, B, Tn)
refs = [ref_of(p.master) for p in ps]
def cv(t, w, b):
    return F.conv2d(t.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=1).permute(0, 2, 3, 1)
y = torch.cat([cv(ref_of(xo), refs[0], refs[1]), cv(ref_of(xf), refs[2], refs[3])], -1).view(B, Tn, H, H, 4)
outr = y.permute(0, 2, 3, 1, 4).reshape(B, H, H, 4 * Tn)
err = (out.detach().double().cpu() - outr).abs()
print('max err', err.max().item())
e = err.view(B, H, H, Tn, 4)
print('by b', e.amax((1,2,3,4)))
print('by t', e.amax((0,1,2,4)))
print('by ch', e.amax((0,1,2,3)))
print('by row', e.amax((0,2,3,4)))
print('by col', e.amax((0,1,3,4)))
