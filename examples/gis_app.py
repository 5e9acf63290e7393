"""Call sequence of the reference's examples/gis_app.py:14-55 against this package: an aerial tile in projected coordinates
(metres, offsets of 1e5 .. 1e6) -- re-centred before reconstruction, sensor-only input with estimated normals, detail_level=0.1,
and mesh.v re-assigned (a numpy array in the caller's coordinate system) before it is written out."""
import numpy as np
import torch
from common import load_las_example, warning_on_low_memory
import nksr

if __name__ == '__main__':
    warning_on_low_memory(4096.0)
    xyz, _ = load_las_example()
    xyz_offset = xyz.mean(0)
    xyz = (xyz - xyz_offset[None]).astype(np.float32)
    xyz = xyz[np.linalg.norm(xyz, axis=1) < 20.0]             # a small region of interest (gis_app.py:28-29)

    device = torch.device("cuda:0")
    reconstructor = nksr.Reconstructor(device)
    reconstructor.chunk_tmp_device = torch.device("cpu")

    input_xyz = torch.from_numpy(xyz).float().to(device)
    input_sensor = torch.tensor([[0.0, 0.0, 50.0]], device=device).repeat(input_xyz.shape[0], 1)

    field = reconstructor.reconstruct(
        input_xyz, sensor=input_sensor, detail_level=0.1,
        approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True,
        preprocess_fn=nksr.get_estimate_normal_preprocess_fn(64, 85.0)
    )
    mesh = field.extract_dual_mesh(mise_iter=1)

    # back to the caller's coordinate system: mesh.v is assignable (gis_app.py:47-52)
    mesh.v = mesh.v.cpu().numpy().astype(float)
    mesh.v += xyz_offset[None, :]
    nksr.utils.write_obj_mesh('gis_app.obj', mesh.v, mesh.f)
    print('V=%d F=%d -> gis_app.obj (coordinates around %s)' % (mesh.v.shape[0], mesh.f.shape[0], np.round(xyz_offset, 1)))
