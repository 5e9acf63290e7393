import time, torch, numpy as np
import nksr_amd
from nksr_amd import utils
from nksr_amd.fields import kernel_field as kf
dev = torch.device('cuda:0')
xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, 0)
x, n = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
rec = nksr_amd.Reconstructor(dev, config='snet-n3k-wnormal')
for i in range(3):
    f = rec.reconstruct(x, n, detail_level=None); m = f.extract_dual_mesh(mise_iter=1)
rec.sync_timing = True
kf.DETAIL = True if hasattr(kf, 'DETAIL') else None
acc = {}
for i in range(5):
    t0 = time.perf_counter()
    f = rec.reconstruct(x, n, detail_level=None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m = f.extract_dual_mesh(mise_iter=1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for k, v in rec.timing.items(): acc[k] = acc.get(k, 0) + v
    acc['reconstruct_total'] = acc.get('reconstruct_total', 0) + t1 - t0
    acc['mesh'] = acc.get('mesh', 0) + t2 - t1
print({k: round(v / 5 * 1e3, 2) for k, v in acc.items()}, f.solve_info['iters'])
print({k: v for k, v in kf.DETAIL_TIMES.items()} if getattr(kf, 'DETAIL_TIMES', None) else None)
