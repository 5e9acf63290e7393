"""Call sequence of the reference's examples/recons_simple.py:16-27 against this package."""
import torch
from common import load_bunny_example, warning_on_low_memory
import nksr

if __name__ == '__main__':
    warning_on_low_memory(1024.0)
    device = torch.device("cuda:0")
    xyz, nrm = load_bunny_example()
    input_xyz = torch.from_numpy(xyz).float().to(device)
    input_normal = torch.from_numpy(nrm).float().to(device)

    reconstructor = nksr.Reconstructor(device)
    field = reconstructor.reconstruct(input_xyz, input_normal, detail_level=1.0)
    mesh = field.extract_dual_mesh(mise_iter=1)

    nksr.utils.write_ply_mesh('recons_simple.ply', mesh.v, mesh.f)
    print('V=%d F=%d -> recons_simple.ply' % (mesh.v.shape[0], mesh.f.shape[0]))
