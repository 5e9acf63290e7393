"""Call sequence of the reference's examples/recons_by_chunk.py:16-30 (chunked reconstruction).
Under torchrun the chunks are sharded over the ranks (one process per GPU)."""
import os
import torch
from common import load_buda_example, warning_on_low_memory
import nksr

if __name__ == '__main__':
    warning_on_low_memory(1024.0 * 7.0)
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get('NKSR_DIST_BACKEND', 'nccl'))
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    xyz, nrm = load_buda_example()
    input_xyz = torch.from_numpy(xyz).float().to(device)
    input_normal = torch.from_numpy(nrm).float().to(device)

    reconstructor = nksr.Reconstructor(device)
    reconstructor.chunk_tmp_device = torch.device("cpu") if world == 1 else device

    field = reconstructor.reconstruct(input_xyz, input_normal, detail_level=None, chunk_size=50.0)
    mesh = field.extract_dual_mesh(mise_iter=1)
    if int(os.environ.get('RANK', 0)) == 0:
        nksr.utils.write_ply_mesh('recons_by_chunk.ply', mesh.v, mesh.f)
        print('chunks=%d V=%d F=%d -> recons_by_chunk.ply' % (len(field.fields), mesh.v.shape[0], mesh.f.shape[0]))
