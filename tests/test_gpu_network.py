"""GPU parity of the sparse feature-hierarchy network (csrc/nn.hip: point MLP, splat-mean, fp32-MFMA
sparse convolution, child pooling, transfer/up-sampling, heads, top-down structure pruning) against
oracle/network.py with the same exported parameters."""
import numpy as np
import pytest
import torch

from conftest import make_cloud
import parity_util as pu

pytestmark = pytest.mark.gpu


def _run(prune):
    import nksr_amd
    from nksr_amd import configs
    from nksr_amd.nn.network import NKSRNetwork
    from oracle import network as onet
    dev = torch.device('cuda:0')
    xyz, nrm = make_cloud('torus', 3000, 0.005, 2)
    xyz = (xyz * np.float32(2.0)).astype(np.float32)
    hp = configs.get_hparams('ks', head_init_scale=0.7, seed=3)
    net = NKSRNetwork(hp)
    rs = np.random.RandomState(0)
    for m in list(net.unet.down) + list(net.unet.up):
        m.bias.data = torch.from_numpy(rs.randn(32).astype(np.float32) * 0.1)
    if prune:   # let the (random) structure head actually decide
        for h in net.unet.structure_heads:
            h.bias.data.zero_()
            h.weight.data *= 4.0
    P = onet.export_params(net)
    net = net.to(dev).eval()          # (inference: in training mode the features carry an autograd graph, nn/backward.py)
    dec_o, basis_o, normals_o, logits_o, trunk_o = onet.forward(P, xyz, nrm, 0.1, 4, 4, 1)
    enc_svh = nksr_amd.SparseFeatureHierarchy(0.1, 4, dev).build_point_splatting(torch.from_numpy(xyz).to(dev))
    enc = net.encoder(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), enc_svh, 0)
    feat, dec, _ = net.unet(enc, enc_svh, adaptive_depth=1)
    return dec_o, basis_o, normals_o, logits_o, trunk_o, feat, dec


@pytest.mark.parametrize('prune', [False, True])
def test_network_matches_oracle(prune):
    dec_o, basis_o, normals_o, logits_o, trunk_o, feat, dec = _run(prune)
    if prune:
        assert any(dec_o.levels[d].n < n for d, n in enumerate([10 ** 9] * 4)) and dec_o.levels[0].n > 0
    for d in range(4):
        assert np.array_equal(dec.level(d).keys.cpu().numpy(), dec_o.levels[d].keys), 'decoder structure differs at level %d' % d
        assert np.array_equal(dec.level(d).nbr.cpu().numpy(), dec_o.levels[d].nbr)
        tg = feat.trunk_features[d].cpu().numpy()
        scale = np.abs(trunk_o[d]).max()
        np.testing.assert_allclose(tg, trunk_o[d], rtol=0, atol=2e-5 * scale)
        np.testing.assert_allclose(feat.structure_features[d].cpu().numpy(), logits_o[d], rtol=0, atol=1e-4)
        np.testing.assert_allclose(feat.basis_features[d].cpu().numpy(), basis_o[d], rtol=0, atol=1e-4)
    # unit normals: the normalisation amplifies fp32 noise by 1/|n_raw|
    from oracle import network as onet
    nn0 = onet.forward.last_normal_norm[0]
    err = np.abs(feat.normal_features[0].cpu().numpy() - normals_o[0]).max(1)
    assert (err <= 2e-5 / np.maximum(nn0, 1e-8) + 1e-5).all()


def test_conv_is_not_transposed():
    """A = I style check with an asymmetric weight: catches row/col swaps of the MFMA fragments."""
    from nksr_amd.nn.network import SparseConv3
    dev = torch.device('cuda:0')
    n = 70
    x = torch.randn(n, 32, device=dev)
    nbr = torch.full((n, 27), -1, dtype=torch.int32, device=dev)
    nbr[:, 13] = torch.arange(n, dtype=torch.int32, device=dev)        # centre tap only
    nbr[:-1, 14] = torch.arange(1, n, dtype=torch.int32, device=dev)   # +z neighbour = next row
    conv = SparseConv3(32).to(dev)
    conv.weight.data.zero_()
    W13 = torch.arange(32 * 32, dtype=torch.float32, device=dev).view(32, 32) / 1000.0   # asymmetric
    conv.weight.data[13] = W13
    conv.weight.data[14] = W13.T * 0.5
    out = conv(x, nbr, relu=False)
    ref = x @ W13
    ref[:-1] += x[1:] @ (W13.T * 0.5)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_reconstruct_with_active_heads_matches_oracle_pipeline():
    """Non-zero residual heads: the network output really feeds the kernel solve on both sides."""
    import nksr_amd
    from nksr_amd import configs
    from oracle import network as onet, pipeline
    dev = torch.device('cuda:0')
    xyz, nrm = make_cloud('sphere', 3000, 0.005, 0)
    hp = configs.get_hparams('ks', head_init_scale=0.15, interpolator_init_scale=0.4, seed=11)
    rec = nksr_amd.Reconstructor(dev, hparams=hp)
    P = onet.export_params(rec.network)
    fld = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), voxel_size=0.05, solver_tol=1e-6)
    xs = (xyz * np.float32(2.0)).astype(np.float32)
    ofl = pipeline.reconstruct(xs, nrm, tol=1e-6, net_params=P)
    assert fld.solve_info['M'] == ofl['A'].shape[0]
    for d in range(4):
        np.testing.assert_allclose(fld._feat[d].cpu().numpy(), ofl['feats'][d], rtol=0, atol=2e-4)
    fo, go = pipeline.evaluate(ofl, xs, grad=True)
    res = fld.evaluate_f(torch.from_numpy(xyz).to(dev), grad=True)
    ref = np.abs(ofl['alpha']).max()
    pu.check('active_heads:f_at_inputs', np.abs(res.value.cpu().numpy() - fo).max() / ref, 1e-4)
    pu.check('active_heads:grad_rel', np.abs(res.gradient.cpu().numpy() / 2.0 - go).max() / np.abs(go).max(), 1e-4)
    pu.mesh_parity('active_heads[mise=1]', fld, ofl, 1, fld.scale)


def test_shapenet_3k_noise_config():
    """BASELINE.json configs[1] stand-in: ShapeNet 3K-noise recipe (sigma=0.005, N=3000,
    configs/shapenet/train_3k_noise.yaml:4-18: voxel_size 0.02, kernel_dim 16, interpolator 2x32) on an
    analytic shape; HIP vs oracle: voxel sets exact, alpha / f within fp32 tolerance."""
    import nksr_amd
    from oracle import network as onet, pipeline
    dev = torch.device('cuda:0')
    xyz, nrm = make_cloud('sphere', 3000, 0.005, 0)
    rec = nksr_amd.Reconstructor(dev, config='snet-n3k-wnormal')
    assert rec.hparams.kernel_dim == 16 and rec.hparams.interpolator.hidden_dim == 32 and rec.hparams.voxel_size == 0.02
    fld = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), detail_level=None, solver_tol=1e-6)
    ofl = pipeline.reconstruct(xyz, nrm, voxel_size=0.02, kernel_dim=16, hidden=32, tol=1e-6,
                               net_params=onet.export_params(rec.network))
    for d in range(4):
        assert np.array_equal(fld.svh.level(d).keys.cpu().numpy(), ofl['hier'].levels[d].keys)
    ref = np.abs(ofl['alpha']).max()
    pu.check_alpha('shapenet3k', fld.alpha.cpu().numpy(), ofl, 1e-6)
    fo, _ = pipeline.evaluate(ofl, xyz)
    fg = fld.evaluate_f(torch.from_numpy(xyz).to(dev)).value.cpu().numpy()
    pu.check('shapenet3k:f_at_inputs', np.abs(fg - fo).max() / ref, 1e-4)
    pu.mesh_parity('shapenet3k[mise=0]', fld, ofl, 0, fld.scale, w0=rec.hparams.voxel_size)


def _open_sheet(n, seed):
    """Open surface (a wavy sheet): the kernel field closes it far from the data; the UDF mask trims that."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1.0, 1.0, (n, 2)).astype(np.float32)
    z = (0.15 * np.sin(2.5 * u[:, 0]) * np.cos(2.0 * u[:, 1])).astype(np.float32)
    xyz = np.stack([u[:, 0], u[:, 1], z], 1).astype(np.float32)
    gx = 0.15 * 2.5 * np.cos(2.5 * u[:, 0]) * np.cos(2.0 * u[:, 1])
    gy = -0.15 * 2.0 * np.sin(2.5 * u[:, 0]) * np.sin(2.0 * u[:, 1])
    nrm = np.stack([-gx, -gy, np.ones(n)], 1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    return xyz, nrm


def test_udf_mask_branch_matches_oracle():
    """udf.enabled (configs/carla/train.yaml:8-9): NeuralField(udf_svh, udf_decoder, udf_features) with
    level set 2*voxel_size (models/nksr_net.py:124-130) against the oracle restatement."""
    import nksr_amd
    from nksr_amd import configs
    from oracle import network as onet, pipeline
    dev = torch.device('cuda:0')
    xyz, nrm = _open_sheet(6000, 0)
    hp = configs.get_hparams('carla')
    assert hp.udf.enabled and hp.adaptive_depth == 2
    rec = nksr_amd.Reconstructor(dev, hparams=hp)
    vs = 0.05
    fld = rec.reconstruct(torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), voxel_size=vs, solver_tol=1e-6)
    assert isinstance(fld.mask_field, nksr_amd.fields.NeuralField) and fld.mask_field.level_set == pytest.approx(0.2)
    xs = (xyz * np.float32(0.1 / vs)).astype(np.float32)
    ofl = pipeline.reconstruct(xs, nrm, adaptive_depth=2, tol=1e-6, net_params=onet.export_params(rec.network), udf=True)
    for d in range(2):
        got = fld.mask_field.features[d].cpu().numpy()
        np.testing.assert_allclose(got, ofl['udf_feats'][d], atol=2e-4)   # fp32 vs fp64 accumulation of a ratio
    # decoded distances: on the data, off the data, far away
    rng = np.random.default_rng(1)
    q = np.concatenate([xs[:2000], xs[:2000] + rng.normal(0, 0.08, (2000, 3)).astype(np.float32),
                        rng.uniform(-3, 3, (2000, 3)).astype(np.float32)]).astype(np.float32)
    got = fld.mask_field._evaluate_f_model(torch.from_numpy(q).to(dev), False).value.cpu().numpy()
    ref = onet.udf_decode(ofl['hier'], ofl['udf_feats'], q)
    far = ref > 1e29
    assert np.array_equal(far, got > 1e29)
    np.testing.assert_allclose(got[~far], ref[~far], atol=2e-5)
    assert np.median(got[:2000]) < 0.02            # on the sheet: distance ~ 0 (model units, voxel = 0.1)
    # meshes: same trimming; the trimmed mesh stays within 2 voxels of the data plane estimate
    _, mesh, _ = pu.mesh_parity('udf[level_set=0.2]', fld, ofl, 0, fld.scale)
    d_v = rec.network.udf_decoder(mesh.v.to(dev) * fld.scale, fld.svh, ofl_feats_to_torch(ofl, dev))
    assert float(d_v.max()) < 0.2 + 1e-6
    # a tight level set actually trims (noise-free sheet: vertices farther than 0.004 from the plane estimate)
    fld.mask_field.set_level_set(0.004)
    ofl['udf_level_set'] = 0.004
    tight = fld.extract_dual_mesh(mise_iter=0)
    tv, tf = pipeline.extract_dual_mesh(ofl, mise_iter=0)
    assert 0 < tight.f.shape[0] < mesh.f.shape[0]
    # the trim threshold (udf < level set) is a second fp32 decision per vertex: triangle sets may differ where a
    # vertex's decoded distance sits at the level set; everything else must be the same triangles
    tg = pu.canonical_triangles(tight.f.cpu().numpy(), tight.edge_vkey.cpu().numpy(), tight.edge_axis.cpu().numpy().astype(np.int64))
    info_t = {}
    tv, tf = pipeline.extract_dual_mesh(ofl, mise_iter=0, info=info_t)
    to = pu.canonical_triangles(tf, info_t['vert_vkey'], info_t['vert_axis'])
    diff = int((~np.isin(pu._rows_view(tg), pu._rows_view(to))).sum() + (~np.isin(pu._rows_view(to), pu._rows_view(tg))).sum())
    pu.report('udf[tight]', T_hip=len(tg), T_oracle=len(to), differing=diff)
    assert diff <= max(8, 0.005 * len(to))
    fld.to_('cpu')                                  # the mask's features travel with the field
    assert fld.mask_field.features[0].device.type == 'cpu'


def ofl_feats_to_torch(ofl, dev):
    return [None if f is None else torch.from_numpy(f).to(dev) for f in ofl['udf_feats']]


def _conv3_torch(x, nbr, W, b):
    n, C = x.shape
    xp = torch.cat([x, x.new_zeros(1, C)])
    idx = torch.where(nbr >= 0, nbr, torch.full_like(nbr, n)).long()
    return torch.relu(b + torch.einsum('nsc,scd->nd', xp[idx], W))


def _torch_forward(net, enc, enc_svh, tape, hp):
    """The network's forward as differentiable torch ops on the SAME discrete structure the HIP forward decided (hierarchies,
    pruning masks, gather tables from ``tape``): what torch autograd differentiates as the reference of nn/backward.py."""
    from nksr_amd.nn.backward import point_corners
    D = enc_svh.depth
    e, u = net.encoder, net.unet
    p0 = enc.xyz * float(enc_svh.inv_w0)
    inp = torch.cat([(p0 - torch.floor(p0)) - 0.5, enc.feat], 1)
    g = torch.relu(inp @ e.W1.t() + e.b1) @ e.W2.t() + e.b2
    idx, w = point_corners(enc_svh.level(0), 0, enc_svh.inv_w0, enc.xyz)
    nv0 = enc_svh.level(0).num_voxels
    ws = torch.zeros(nv0, device=g.device).index_add_(0, idx.reshape(-1), w.reshape(-1))
    acc = torch.zeros(nv0, g.shape[1], device=g.device).index_add_(0, idx.reshape(-1), (w[..., None] * g[:, None, :]).reshape(-1, g.shape[1]))
    vf = acc / ws.clamp_min(1e-30)[:, None] * (ws > 0)[:, None]
    x = [_conv3_torch(vf, enc_svh.level(0).nbr, u.down[0].weight, u.down[0].bias)]
    for d in range(1, D):
        st, en = tape['ranges'][d]
        nchild = x[d - 1].shape[0]
        kid = torch.arange(nchild, device=g.device)
        par = torch.searchsorted(en.long(), kid, right=True).clamp_max(en.numel() - 1)
        inside = ((st.long()[par] <= kid) & (kid < en.long()[par])).to(torch.float32)
        cnt = (en - st).to(torch.float32).clamp_min(1.0)
        pool = torch.zeros(en.numel(), x[d - 1].shape[1], device=g.device).index_add_(0, par, x[d - 1] * inside[:, None]) / cnt[:, None]
        x.append(_conv3_torch(pool, enc_svh.level(d).nbr, u.down[d].weight, u.down[d].bias))
    out = {}
    y_up = None
    trunk = [None] * D
    for d in range(D - 1, -1, -1):
        rec = tape['dec'][d]
        je = rec['je'].long()
        t = x[d][je.clamp_min(0)] * (je >= 0)[:, None]
        if rec['par'] is not None:
            t = t + y_up[rec['par'].long()]
        y = _conv3_torch(t, rec['nbr'], u.up[d].weight, u.up[d].bias)
        if rec['exist'] is not None:
            y = y[rec['exist']]
        trunk[d] = y
        y_up = y
    e0 = torch.zeros(int(hp.kernel_dim), device=g.device)
    e0[0] = 1.0
    for d in range(D):
        y = trunk[d]
        out[('basis', d)] = y @ u.basis_heads[d].weight.t() + u.basis_heads[d].bias + e0
        out[('structure', d)] = y @ u.structure_heads[d].weight.t() + u.structure_heads[d].bias
        if tape['normal'][d] is not None:
            nv_hip, den_hip, by_norm = tape['normal'][d]
            head_hip = (tape['trunk'][d] @ u.normal_heads[d].weight.t() + u.normal_heads[d].bias).detach()
            nv = (nv_hip - head_hip) + y @ u.normal_heads[d].weight.t() + u.normal_heads[d].bias
            den = torch.where(by_norm, nv.norm(dim=1), den_hip)
            out[('normal', d)] = nv / den[:, None]
        if tape['feat'].udf_features[d] is not None:
            head_hip = (tape['trunk'][d] @ u.udf_heads[d].weight.t() + u.udf_heads[d].bias).detach()
            out[('udf', d)] = (tape['feat'].udf_features[d].detach() - head_hip) + y @ u.udf_heads[d].weight.t() + u.udf_heads[d].bias
    return out


@pytest.mark.parametrize('preset', ['ks', 'carla'])
def test_network_backward_matches_autograd_through_the_torch_forward(preset):
    """models/nksr_net.py:73-78 under autograd: in training mode ``network.encoder`` / ``network.unet`` return features tied to the
    reverse sweep of nn/backward.py (sparse-convolution data gradients through the MFMA kernel with mirrored taps, weight gradients as
    library GEMMs, transposed pooling / gathers / splats).  Checked against torch autograd through a torch statement of the same
    forward on the same discrete structure: every parameter of the encoder and of the U-Net (pruning active, both presets: one /
    two levels of normal targets, UDF heads), relative L2 per parameter."""
    import nksr_amd
    from nksr_amd import configs
    from nksr_amd.nn.network import NKSRNetwork
    dev = torch.device('cuda:0')
    xyz, nrm = make_cloud('torus', 3000, 0.005, 2)
    xyz = (xyz * np.float32(2.0)).astype(np.float32)
    hp = configs.get_hparams(preset, head_init_scale=0.7, seed=3)
    net = NKSRNetwork(hp)
    rs = np.random.RandomState(0)
    for m in list(net.unet.down) + list(net.unet.up):
        m.bias.data = torch.from_numpy(rs.randn(32).astype(np.float32) * 0.1)
    for h in net.unet.structure_heads:        # let the (random) structure head prune
        h.bias.data.zero_()
        h.weight.data *= 4.0
    net = net.to(dev).train()
    t = lambda a: torch.from_numpy(a).to(dev)
    enc_svh = nksr_amd.SparseFeatureHierarchy(0.1, 4, dev).build_point_splatting(t(xyz))
    ad = int(hp.adaptive_depth)
    enc = net.encoder(t(xyz), t(nrm), enc_svh, 0)
    assert enc.voxel_feat.requires_grad
    feat, dec, _ = net.unet(enc, enc_svh, adaptive_depth=ad)
    assert any(dec.level(d).num_voxels < enc_svh.level(d).num_voxels * 27 for d in range(4))
    outs = {('basis', d): feat.basis_features[d] for d in range(4)}
    outs.update({('structure', d): feat.structure_features[d] for d in range(4)})
    outs.update({('normal', d): feat.normal_features[d] for d in range(4) if feat.normal_features[d] is not None})
    outs.update({('udf', d): feat.udf_features[d] for d in range(4) if feat.udf_features[d] is not None})
    assert all(o.requires_grad for o in outs.values()) and len([k for k in outs if k[0] == 'normal']) == ad
    gen = torch.Generator(device='cpu').manual_seed(5)
    coef = {k: torch.randn(o.shape, generator=gen).to(dev) for k, o in sorted(outs.items())}
    params = [q for q in net.encoder.parameters()] + [q for q in net.unet.parameters()]
    names = ['encoder.' + n for n, _ in net.encoder.named_parameters()] + ['unet.' + n for n, _ in net.unet.named_parameters()]
    loss = sum((coef[k] * o).sum() for k, o in outs.items())
    got = torch.autograd.grad(loss, params, allow_unused=True)
    # the reference: torch autograd through the torch statement of the forward, same structure
    from nksr_amd.nn.backward import UNetFunction  # noqa: F401
    tape = {}
    with torch.no_grad():
        enc2 = net.encoder(t(xyz), t(nrm), enc_svh, 0)
        feat2, dec2, _ = net.unet._forward_impl(enc2, enc_svh, ad, None, tape)
    tape['feat'] = feat2
    ref_out = _torch_forward(net, enc2, enc_svh, tape, hp)
    for k, o in outs.items():
        scale = float(ref_out[k].detach().abs().max()) + 1e-12
        pu.check('network_backward[%s]:forward_%s_%d' % (preset, k[0], k[1]), float((ref_out[k] - o).abs().max()) / scale, 2e-4)
    loss_ref = sum((coef[k] * ref_out[k]).sum() for k in outs)
    ref = torch.autograd.grad(loss_ref, params, allow_unused=True)
    worst = 0.0
    for n_, a, b in zip(names, got, ref):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) < 1e-6, n_
            continue
        rel = float((a.double() - b.double()).norm() / b.double().norm())
        worst = max(worst, rel)
        assert rel < 2e-3, (n_, rel)
    pu.check('network_backward[%s]:worst_parameter_rel_l2' % preset, worst, 2e-3)


def test_training_step_end_to_end_gradients_reach_every_parameter():
    """One training step the way models/nksr_net.py:57-112 runs it: point encoder -> U-Net (training mode) -> KernelField on the
    predicted basis features and normal targets -> solve_non_fused -> evaluate_f at the input points -> loss -> backward.  The
    gradient reaches the interpolators (theta term, HIP), the normal targets (adjoint solve + gradient evaluation, HIP) and through
    them every parameter of the U-Net and the encoder (nn/backward.py); the directional derivative along a random direction in
    parameter space matches a central difference of the same HIP forward (structure kept fixed by the analytic structure head)."""
    import nksr_amd
    from nksr_amd import configs
    from nksr_amd.fields import KernelField
    from nksr_amd.nn.network import NKSRNetwork
    dev = torch.device('cuda:0')
    xyz, nrm = make_cloud('sphere', 1500, 0.003, 4)
    xyz = (xyz * np.float32(2.0)).astype(np.float32)
    hp = configs.get_hparams('ks', seed=1, interpolator_init_scale=0.3)
    net = NKSRNetwork(hp)
    gen = torch.Generator().manual_seed(9)
    for hs in (net.unet.basis_heads, net.unet.normal_heads):          # non-trivial heads; the structure head stays analytic (no pruning flips)
        for h in hs:
            h.weight.data = 0.2 * torch.randn(h.weight.shape, generator=gen) / h.cin ** 0.5
    net = net.to(dev).train()
    t = lambda a: torch.from_numpy(a).to(dev)
    X, N = t(xyz), t(nrm)

    def loss_fn():
        enc_svh = nksr_amd.SparseFeatureHierarchy(0.1, 4, dev).build_point_splatting(X)
        enc = net.encoder(X, N, enc_svh, 0)
        feat, dec, _ = net.unet(enc, enc_svh, adaptive_depth=1)
        # (approx_kernel_grad: the exact-gradient rows are piecewise constant in theta across ReLU kinks -- differences of the loss are
        # only a valid check of the derivative where the rows are continuous, tests/test_gpu_parity.py says the same of the theta term)
        fld = KernelField(svh=dec, interpolator=net.interpolators, features=feat.basis_features, approx_kernel_grad=True)
        fld.solver_config.update({'tol': 1e-7, 'max_iter': 4000})
        nxyz = dec.get_voxel_centers(0)
        fld.solve_non_fused(pos_xyz=enc.xyz, normal_xyz=nxyz, normal_value=-feat.normal_features[0], pos_weight=1e4 / X.shape[0],
                            normal_weight=1e4 / nxyz.shape[0] * 0.01, reg_weight=1.0)
        res = fld.evaluate_f(enc.xyz + 0.02 * enc.feat, grad=True)
        return 1e3 * ((res.value ** 2).mean() + ((res.gradient + enc.feat) ** 2).sum(1).mean() * 0.01)
    params = [q for q in net.parameters() if q.requires_grad]
    loss = loss_fn()
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    names = [n for n, q in net.named_parameters() if q.requires_grad]
    reached = {n: (g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0) for n, g in zip(names, grads)}
    for group in ('encoder.', 'unet.down.', 'unet.up.', 'unet.basis_heads.0', 'unet.normal_heads.0', 'interpolators.0'):
        assert any(ok for n, ok in reached.items() if n.startswith(group)), (group, [n for n in reached if n.startswith(group)])
    # directional derivative vs central difference of the HIP forward
    gen = torch.Generator().manual_seed(3)
    sel = [(q, g) for n, q, g in zip(names, params, grads) if g is not None and not n.startswith('unet.structure') and not n.startswith('unet.udf')]
    v = [torch.randn(q.shape, generator=gen).to(dev) * q.detach().abs().mean().clamp_min(1e-3) for q, _ in sel]
    analytic = sum(float((g * d).sum()) for (_, g), d in zip(sel, v))
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sign in (1.0, -1.0):
            for (q, _), d in zip(sel, v):
                q.add_(sign * eps * d)
            vals.append(float(loss_fn()))
            for (q, _), d in zip(sel, v):
                q.sub_(sign * eps * d)
    fd = (vals[0] - vals[1]) / (2 * eps)
    pu.report('training_step:directional_derivative', analytic=analytic, finite_difference=fd, loss=float(loss))
    # (a coarse end-to-end guard -- a missing path shows as an O(1) error; the U-Net's ReLUs and the target normalisation put kinks
    # between +-eps, the precise pins are the per-stage tests: theta term 1e-7, network sweep 1e-6 of their autograd references)
    pu.check('training_step:directional_derivative_rel', abs(analytic - fd) / max(abs(fd), 1e-12), 1e-1)


def test_conv_weight_gradient_kernel_matches_the_gathered_gemm():
    """csrc/nn.hip k_conv3_wgrad (fp32 MFMA, two voxels per instruction, per-chunk partials) against the 27 GEMMs over the gathered
    taps in torch, on a grid larger than one chunk and with absent neighbours; two runs are bit-identical (no atomics)."""
    import nksr_amd
    from nksr_amd.nn.backward import conv3_wgrad
    dev = torch.device('cuda:0')
    xyz, _ = make_cloud('torus', 20000, 0.0, 1)
    svh = nksr_amd.SparseFeatureHierarchy(0.1, 1, dev).build_point_splatting(torch.from_numpy((xyz * np.float32(6.0)).astype(np.float32)).to(dev))
    g = svh.level(0)
    n = g.num_voxels
    assert n > 3 * 2048 and bool((g.nbr < 0).any())
    gen = torch.Generator().manual_seed(0)
    x, gz = torch.randn(n, 32, generator=gen).to(dev), torch.randn(n, 32, generator=gen).to(dev)
    got = conv3_wgrad(x, g.nbr, gz)
    xp = torch.cat([x, x.new_zeros(1, 32)]).double()
    idx = torch.where(g.nbr >= 0, g.nbr, torch.full_like(g.nbr, n)).long()
    ref = torch.zeros(27, 32, 32, dtype=torch.float64, device=dev)
    for s0 in range(0, n, 8192):
        ref += torch.einsum('nsc,nd->scd', xp[idx[s0:s0 + 8192]], gz[s0:s0 + 8192].double())
    pu.check('conv3_wgrad:rel_l2', float((got.double() - ref).norm() / ref.norm()), 1e-5)
    assert torch.equal(got, conv3_wgrad(x, g.nbr, gz))
