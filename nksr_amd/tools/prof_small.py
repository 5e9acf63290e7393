import cProfile
import pstats
import time

import torch

import nksr_amd
from nksr_amd import utils
dev = torch.device('cuda:0')
xyz, nrm = utils.synth_scene(150000, seed=0)
xyz = torch.from_numpy(xyz).to(dev); nrm = torch.from_numpy(nrm).to(dev)
rec = nksr_amd.Reconstructor(dev)
for i in range(3):
    f = rec.reconstruct(xyz, nrm, detail_level=1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5):
    f = rec.reconstruct(xyz, nrm, detail_level=1.0)
torch.cuda.synchronize()
print('warm per-call ms', (time.perf_counter() - t0) / 5 * 1e3, f.solve_info)
rec.sync_timing = True
f = rec.reconstruct(xyz, nrm, detail_level=1.0)
print({k: round(v * 1e3, 2) for k, v in rec.timing.items()})
rec.sync_timing = False
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    f = rec.reconstruct(xyz, nrm, detail_level=1.0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
