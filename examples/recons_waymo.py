"""Call sequence of the reference's examples/recons_waymo.py:19-43 / recons_waymo_cpu.py:44-63:
sensor-only input, normals estimated by the GPU kNN-PCA preprocess."""
import torch
from common import load_waymo_example, warning_on_low_memory
import nksr

if __name__ == '__main__':
    warning_on_low_memory(20000.0)
    xyz_np, sensor_np = load_waymo_example()
    device = torch.device("cuda:0")
    reconstructor = nksr.Reconstructor(device)
    reconstructor.chunk_tmp_device = torch.device("cpu")

    input_xyz = torch.from_numpy(xyz_np).float().to(device)
    input_sensor = torch.from_numpy(sensor_np).float().to(device)

    field = reconstructor.reconstruct(
        input_xyz, sensor=input_sensor, detail_level=None,
        # Minor configs for better efficiency (not necessary)
        approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True,
        chunk_size=51.2,   # (commented out in the reference script; the synthetic stand-in needs it)
        preprocess_fn=nksr.get_estimate_normal_preprocess_fn(64, 85.0)
    )
    mesh = field.extract_dual_mesh(mise_iter=1)
    nksr.utils.write_ply_mesh('recons_waymo.ply', mesh.v, mesh.f)
    print('V=%d F=%d -> recons_waymo.ply' % (mesh.v.shape[0], mesh.f.shape[0]))
