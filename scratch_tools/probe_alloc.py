import sys, time, torch
sys.path.insert(0, '/root/repo')
import nksr_amd
from nksr_amd import utils
dev = torch.device('cuda:0')
xyz, nrm = utils.synth_scene(1_000_000, seed=0)
xyz = torch.from_numpy(xyz).to(dev); nrm = torch.from_numpy(nrm).to(dev)
rec = nksr_amd.Reconstructor(dev); rec.sync_timing = True
for it in range(4):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = rec.reconstruct(xyz, nrm, detail_level=1.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m = f.extract_dual_mesh(mise_iter=1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print(it, 'recon %.1f ms mesh %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), rec.timing,
          'dev_alloc', s1['num_device_alloc'] - s0['num_device_alloc'], 'dev_free', s1['num_device_free'] - s0['num_device_free'],
          'retries', s1['num_alloc_retries'] - s0['num_alloc_retries'], 'reserved GB %.1f' % (s1['reserved_bytes.all.current'] / 2**30),
          'peak alloc GB %.1f' % (s1['allocated_bytes.all.peak'] / 2**30))
    del f, m
