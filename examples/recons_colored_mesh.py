"""Call sequence of the reference's examples/recons_colored_mesh.py:20-31."""
import torch
from common import load_spot_example, warning_on_low_memory
from nksr import Reconstructor, utils, fields

if __name__ == '__main__':
    warning_on_low_memory(1024.0)
    device = torch.device("cuda:0")
    xyz, nrm, col = load_spot_example()
    input_xyz = torch.from_numpy(xyz).float().to(device)
    input_normal = torch.from_numpy(nrm).float().to(device)
    input_color = torch.from_numpy(col).float().to(device)

    reconstructor = Reconstructor(device)
    field = reconstructor.reconstruct(input_xyz, input_normal, detail_level=1.0)
    field.set_texture_field(fields.PCNNField(input_xyz, input_color))
    mesh = field.extract_dual_mesh(max_points=2 ** 22, mise_iter=1)

    utils.write_ply_mesh('recons_colored.ply', mesh.v, mesh.f, mesh.c)
    print('V=%d F=%d (coloured) -> recons_colored.ply' % (mesh.v.shape[0], mesh.f.shape[0]))
