"""Call sequence of the reference's examples/recons_scannet.py:19-29 against this package: an indoor scan with a fixed finest
voxel (voxel_size=0.02 instead of a detail level) and two rounds of MISE refinement."""
import torch
from common import load_scannet_example, warning_on_low_memory
import nksr

if __name__ == '__main__':
    warning_on_low_memory(4096.0)
    device = torch.device("cuda:0")
    xyz, nrm = load_scannet_example()
    input_xyz = torch.from_numpy(xyz).float().to(device)
    input_normal = torch.from_numpy(nrm).float().to(device)

    reconstructor = nksr.Reconstructor(device)
    field = reconstructor.reconstruct(input_xyz, input_normal, voxel_size=0.02)
    mesh = field.extract_dual_mesh(mise_iter=2)

    nksr.utils.write_ply_mesh('recons_scannet.ply', mesh.v, mesh.f)
    print('V=%d F=%d -> recons_scannet.ply' % (mesh.v.shape[0], mesh.f.shape[0]))
