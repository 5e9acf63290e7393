"""A training step on the MI355X with the pieces of this package, following the reference's model forward
(models/nksr_net.py:57-112: encoder hierarchy -> network.encoder -> network.unet -> KernelField -> solve_non_fused) and its
losses on the predicted field (models/loss.py:189-198: field value / gradient at samples; :152-160: structure cross-entropy).
The ground truth here is analytic (a sphere) instead of the datasets of the reference's training framework, which is out of scope;
`ext.sdfgen.sdf_from_points` makes the SDF supervision the way the reference does (models/loss.py:85)."""
import numpy as np
import torch

import common  # noqa: F401  (puts the repository on sys.path)
import ext
import nksr
from nksr import SparseFeatureHierarchy
from nksr.fields import KernelField

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(0)
    v = rs.randn(4000, 3)
    nrm = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    xyz = (nrm * 0.9 + rs.randn(4000, 3) * 0.004).astype(np.float32)
    X, N = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    hp = nksr.configs.get_hparams('ks', interpolator_init_scale=0.1, head_init_scale=0.1)
    net = nksr.NKSRNetwork(hp).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    q = torch.from_numpy((xyz[::4] + nrm[::4] * rs.randn(1000, 1).astype(np.float32) * 0.05).astype(np.float32)).to(dev)
    gt_sdf = ext.sdfgen.sdf_from_points(q, X, N, nb_points=8, stdv=0.02)[0]             # training ground truth (models/loss.py:85)
    for step in range(5):
        enc_svh = SparseFeatureHierarchy(voxel_size=hp.voxel_size, depth=hp.tree_depth, device=dev)
        enc_svh.build_point_splatting(X)
        enc = net.encoder(X, N, enc_svh, 0)
        feat, dec_svh, _ = net.unet(enc, enc_svh, adaptive_depth=hp.adaptive_depth)
        field = KernelField(svh=dec_svh, interpolator=net.interpolators, features=feat.basis_features, approx_kernel_grad=True)
        nxyz = dec_svh.get_voxel_centers(0)
        field.solve_non_fused(pos_xyz=enc.xyz, normal_xyz=nxyz, normal_value=-feat.normal_features[0],
                              pos_weight=hp.solver.pos_weight / X.shape[0],
                              normal_weight=hp.solver.normal_weight / nxyz.shape[0] * hp.voxel_size ** 2, reg_weight=1.0)
        res = field.evaluate_f(q, grad=True)
        # f > 0 inside, the reference's target is the negated signed distance (models/loss.py:85,99-100)
        loss = ((res.value + gt_sdf) ** 2).mean() + 0.01 * ((res.gradient.norm(dim=1) - 1.0) ** 2).mean()
        gt_status = dec_svh.evaluate_voxel_status(dec_svh.grids[0], 0)                    # (self-consistent stand-in for the GT hierarchy)
        loss = loss + 0.01 * torch.nn.functional.cross_entropy(feat.structure_features[0], gt_status)
        opt.zero_grad()
        loss.backward()
        gnorm = float(torch.sqrt(sum((p.grad ** 2).sum() for p in net.parameters() if p.grad is not None)))
        opt.step()
        print('step %d  loss %.5f  |grad| %.4f  PCG iters %d  unknowns %d' % (step, float(loss.detach()), gnorm, field.solve_info['iters'], field.solve_info['M']))
