import os, sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np, torch
import test_gpu_parity as T
from nksr_amd.fields import KernelField
xyz, nrm, oh, svh, feats, ointerps, net = T._setup(n=3000, init_scale=0.3, H=16)
fld = KernelField(svh, net.interpolators, [torch.from_numpy(f) for f in feats], approx_kernel_grad=False)
dev = torch.device('cuda:0')
t = lambda x: torch.from_numpy(x).to(dev)
nxyz = np.concatenate([oh.levels[0].centers(), oh.levels[1].centers()])
nval = np.random.RandomState(5).randn(len(nxyz), 3).astype(np.float32)
out = {}
for layout in ('dense', 'compact'):
    os.environ['NKSR_ROWS_LAYOUT'] = layout
    op = fld.fused_operator(t(xyz), t(nxyz), t(nval), 1e4 / 3000, 1e2 / len(nxyz))
    out[layout] = (fld.dense_rows(op).clone(), op)
a, b = out['dense'][0], out['compact'][0]
op = out['compact'][1]
L, R = a.shape[0], a.shape[1]
print('rows', R, 'words', op['rows_words'], 'dense', a.numel())
for d in range(L):
    bad = (a[d].view(torch.int32) != b[d].view(torch.int32))
    print('level', d, 'mismatching entries', int(bad.sum()), 'rows with mismatch', int(bad.any(1).sum()), 'of', R)
    if bad.any():
        rows = torch.nonzero(bad.any(1)).reshape(-1)[:6].tolist()
        for r in rows:
            c = int(op['row_cells'][d, r]); tb = op['nbr32'][c].tolist()
            print('  row', r, 'cell', c, 'first', tb[28], 'last', tb[29], 'b4', tb[30], 'mask %x' % tb[31], 'k', bin(tb[31]).count('1'), 'r-first', r - tb[28], 'r%64', r % 64)
            print('    dense  ', [round(v, 4) for v in a[d, r].tolist()])
            print('    compact', [round(v, 4) for v in b[d, r].tolist()])
# raw check of monotone layout
tb = op['nbr32'].long()
print('zero block', op['rows_all'][:4].tolist())
d = 0
bad = (a[d].view(torch.int32) != b[d].view(torch.int32))
rows = torch.nonzero(bad.any(1)).reshape(-1)[:2].tolist()
for r in rows:
    v = b[d, r, :4]
    for dd in range(L):
        for s0 in range(0, 24):
            m = (a[dd][:, s0:s0 + 4] == v[None]).all(1)
            if m.any():
                print('row', r, 'first 4 compact words == dense level', dd, 'row', torch.nonzero(m).reshape(-1).tolist()[:4], 'slots from', s0)
    c = int(op['row_cells'][d, r]); tb = op['nbr32'][c].tolist()
    k = bin(tb[31]).count('1')
    g = tb[30] * 4 + (r - tb[28]) * k
    print('raw words around g=%d:' % g, [round(x, 4) for x in op['rows_all'][g - 6:g + 8].tolist()])
    print('dense prev row tail', [round(x, 4) for x in a[d, r - 1, -6:].tolist()], 'this row head', [round(x, 4) for x in a[d, r, :8].tolist()])
