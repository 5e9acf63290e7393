"""BASELINE.json configs[4] stand-in: synthetic 10M-point km-scale terrain, tree_depth=5, 8x8 chunks of
125 m.  Chunk mode takes a pre-scaled cloud (NKSR-USAGE.md:137): the
scene is scaled so that an occupied finest voxel holds ~4 points.  Under torchrun the 64 chunks are
sharded over the ranks."""
import os
import sys
import time

import torch
import common  # noqa: F401  (puts the repository on sys.path)
import nksr
from nksr_amd import configs, utils
from nksr_amd.density import scale_for_detail_level

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get('NKSR_DIST_BACKEND', 'nccl'))
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    xyz_np, nrm_np = utils.synth_terrain(n, seed=0, extent=(1000.0, 1000.0))
    xyz = torch.from_numpy(xyz_np).to(device)
    nrm = torch.from_numpy(nrm_np).to(device)
    rec = nksr.Reconstructor(device, hparams=configs.get_hparams('ks', tree_depth=5))
    scale = scale_for_detail_level(xyz[:1_000_000].contiguous(), 1.0, rec.hparams.voxel_size) if n >= 1_000_000 else 1.0
    # the 1M-point prefix is a uniform sample of the whole square: its density is n/1e6 lower
    scale *= (n / min(n, 1_000_000)) ** 0.5
    xs = (xyz * scale).contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    field = rec.reconstruct(xs, nrm, detail_level=None, chunk_size=125.0 * scale)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mesh = field.extract_dual_mesh(mise_iter=1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if int(os.environ.get('RANK', 0)) == 0:
        print('N=%d scale=%.4f chunks=%d (grid %s) solve %.2fs mesh %.2fs -> %.2f Mpts/s, V=%d F=%d' % (
            n, scale, len(field.fields), field.grid, t1 - t0, t2 - t1, n / (t2 - t0) / 1e6, mesh.v.shape[0], mesh.f.shape[0]))
