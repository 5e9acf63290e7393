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
far = (xyz[:40] + np.float32(3.0)).astype(np.float32)
xv = torch.randn(svh.num_unknowns, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
for pos, ns in ((np.concatenate([xyz, far]), nxyz), (xyz, None), (None, nxyz)):
    out = {}
    for layout in ('dense', 'compact', 'dense2'):
        os.environ['NKSR_ROWS_LAYOUT'] = layout.rstrip('2')
        os.environ['NKSR_ROWS_KERNEL'] = 'site' if layout == 'dense2' else 'merged'
        op = fld.fused_operator(t(pos) if pos is not None else None, t(ns) if ns is not None else None, t(nval) if ns is not None else None, 1e4 / 3000, 1e2 / len(nxyz))
        b, dg = fld.fused_rhs_diag(op, 1.0)
        out[layout] = (b, dg, fld.fused_apply(op, xv), fld.fused_apply(op, xv))
    for k in ('compact', 'dense2'):
        for i, nm in enumerate(('rhs', 'diag', 'Ax', 'Ax again')):
            a_, b_ = out['dense'][i], out[k][i]
            nd = int((a_ != b_).sum())
            print(k, nm, 'nan', int(torch.isnan(b_).sum()), 'differing', nd, 'max rel', float(((a_ - b_).abs() / a_.abs().clamp(min=1e-20)).max()) if nd else 0.0,
                  'where', torch.nonzero(a_ != b_).reshape(-1)[:5].tolist(), 'offsets', svh.offsets)
