"""NKSRNetwork -- host-side mirror of ``nksr.NKSRNetwork``.

Reference interface (call sites): ``NKSRNetwork(hparams)`` is an ``nn.Module`` with members
``encoder``, ``unet``, ``interpolators``, ``sdf_decoder``, ``udf_decoder``
(models/nksr_net.py:35-38,73-78,93,117,127); ``encoder(xyz, feat, enc_svh, 0)`` :73;
``unet(feat, enc_svh, adaptive_depth=, gt_decoder_svh=) -> (feat, dec_svh, udf_svh)`` :74-78
with ``feat.basis_features[d]``, ``feat.normal_features[d]``, ``feat.structure_features``,
``feat.udf_features`` (:94,101,118,128,136-138).  Hyper-parameters by name:
configs/default/train.yaml:9-29.

The pretrained weights are fetched from the network at run time by the reference
(models/nksr_net.py:36-38) and are unavailable offline, so the default initialisation here is
*analytic + seeded*: heads are residual around quantities that make an untrained network a
sound (non-learned) kernel solver --
  * basis features   = e_0 + unet head         (pure quadratic B-spline kernel when head = 0)
  * normal features  = normalize(splat(input normals) + unet head)
  * structure        = the containing cell + 26 neighbours of every point exist
A user-supplied ``state_dict`` replaces the seeded part.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import ops
from .._lib import call, ptr, stream
from ..svh import SparseFeatureHierarchy


class Interpolator(nn.Module):
    """Per-level MLP  K -> H -> H -> K (ReLU) with a residual skip: phi = t + MLP(t)."""

    def __init__(self, kernel_dim, hidden_dim, n_hidden=2, init_scale=0.0, generator=None):
        super().__init__()
        if n_hidden != 2:
            raise RuntimeError('interpolator.n_hidden must be 2 (configs/default/train.yaml:24)')
        self.kernel_dim, self.hidden_dim = int(kernel_dim), int(hidden_dim)
        K, H = self.kernel_dim, self.hidden_dim

        self.W1 = nn.Parameter(torch.randn(H, K, generator=generator) / K ** 0.5)
        self.b1 = nn.Parameter(torch.zeros(H))
        self.W2 = nn.Parameter(torch.randn(H, H, generator=generator) / H ** 0.5)
        self.b2 = nn.Parameter(torch.zeros(H))
        # zero last layer when init_scale == 0  =>  phi == t (pure B-spline kernel)
        self.W3 = nn.Parameter(float(init_scale) * torch.randn(K, H, generator=generator) / H ** 0.5)
        self.b3 = nn.Parameter(torch.zeros(K))

    def packed(self):
        return torch.cat([p.reshape(-1) for p in (self.W1, self.b1, self.W2, self.b2, self.W3, self.b3)])

    def forward(self, t):
        h = torch.relu(t @ self.W1.T + self.b1)
        h = torch.relu(h @ self.W2.T + self.b2)
        return t + h @ self.W3.T + self.b3


class FeatureSet:
    """What ``unet`` returns as ``feat`` (models/nksr_net.py:94,101,118,128,136-138)."""

    def __init__(self, depth):
        self.basis_features = [None] * depth
        self.normal_features = [None] * depth
        self.structure_features = {}
        self.udf_features = [None] * depth
        self.encoder_features = None


class EncodedCloud:
    """Output of ``network.encoder``: Morton-sorted cloud + per-level site ranges + the
    trilinear splat of the input feature onto the encoder hierarchy's finest level."""

    def __init__(self):
        self.xyz = None
        self.feat = None
        self.keys = None
        self.splat = None
        self.wsum = None


def sort_cloud(xyz, feat, inv_w0):
    n = xyz.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device)
    call('nksr_point_keys', ptr(xyz), n, inv_w0, ptr(keys), stream())
    ks, perm = ops.sort_pairs(keys, torch.arange(n, dtype=torch.int32, device=xyz.device))
    perm = perm.long()
    return ks, xyz[perm].contiguous(), (feat[perm].contiguous() if feat is not None else None)


def splat_trilinear(svh, d, site_keys, xyz_sorted, feat_sorted):
    """Weighted sum and weight sum of ``feat`` splatted onto level d of ``svh``."""
    g = svh.level(d)
    n, C_ = g.num_voxels, feat_sorted.shape[1]
    st = torch.empty(n, dtype=torch.int32, device=svh.device)
    en = torch.empty(n, dtype=torch.int32, device=svh.device)
    call('nksr_site_ranges', ptr(site_keys), site_keys.numel(), ptr(g.keys), n, d, ptr(st), ptr(en), stream())
    out = torch.empty((n, C_), dtype=torch.float32, device=svh.device)
    ws = torch.empty(n, dtype=torch.float32, device=svh.device)
    inv_w = svh.inv_w0 * (2.0 ** (-d))
    call('nksr_splat_trilinear', ptr(xyz_sorted), ptr(feat_sorted), C_, ptr(st), ptr(en), ptr(g.nbr), ptr(g.ijk), n,
         float(inv_w), ptr(out), ptr(ws), stream())
    return out, ws


class PointEncoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams

    def forward(self, xyz, feat, svh, depth=0):
        enc = EncodedCloud()
        if feat is None:
            raise RuntimeError("this network needs an orientation feature (hparams.feature='normal')")
        enc.keys, enc.xyz, enc.feat = sort_cloud(xyz.contiguous(), feat.to(torch.float32).contiguous(), svh.inv_w0)
        return enc


class StructureUNet(nn.Module):
    """Decoder-side structure + heads.  Round-1 scope: the analytic branch (see module doc);
    the sparse-convolution trunk plugs in as residual heads."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams

    def forward(self, enc, enc_svh, adaptive_depth=1, gt_decoder_svh=None):
        hp = self.hparams
        depth = enc_svh.depth
        dev = enc_svh.device
        if gt_decoder_svh is not None:
            dec_svh = gt_decoder_svh
        else:
            dec_svh = SparseFeatureHierarchy(enc_svh.voxel_size, depth, dev).build_point_neighborhood(enc.xyz)
        feat = FeatureSet(depth)
        K = int(hp.kernel_dim)
        for d in range(depth):
            n = dec_svh.num_voxels(d)
            b = torch.zeros((n, K), dtype=torch.float32, device=dev)
            b[:, 0] = 1.0
            feat.basis_features[d] = b
            feat.structure_features[d] = torch.zeros((n, 3), dtype=torch.float32, device=dev)
            if d < adaptive_depth:
                s, _ = splat_trilinear(dec_svh, d, enc.keys, enc.xyz, enc.feat)
                feat.normal_features[d] = s / s.norm(dim=1, keepdim=True).clamp_min(1e-8)
        return feat, dec_svh, dec_svh


class NKSRNetwork(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        gen = torch.Generator().manual_seed(int(getattr(hparams, 'seed', 0)))
        self.encoder = PointEncoder(hparams)
        self.unet = StructureUNet(hparams)
        self.interpolators = nn.ModuleList([
            Interpolator(hparams.kernel_dim, hparams.interpolator.hidden_dim, hparams.interpolator.n_hidden,
                         init_scale=float(getattr(hparams, 'interpolator_init_scale', 0.0)), generator=gen)
            for _ in range(hparams.tree_depth)])
        self.sdf_decoder = None
        self.udf_decoder = None
