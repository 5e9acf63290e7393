"""NKSRNetwork -- host-side mirror of ``nksr.NKSRNetwork``.

Reference interface (call sites): ``NKSRNetwork(hparams)`` is an ``nn.Module`` with members
``encoder``, ``unet``, ``interpolators``, ``sdf_decoder``, ``udf_decoder``
(models/nksr_net.py:35-38,73-78,93,117,127); ``encoder(xyz, feat, enc_svh, 0)`` :73;
``unet(feat, enc_svh, adaptive_depth=, gt_decoder_svh=) -> (feat, dec_svh, udf_svh)`` :74-78
with ``feat.basis_features[d]``, ``feat.normal_features[d]``, ``feat.structure_features``,
``feat.udf_features`` (:94,101,118,128,136-138).  Hyper-parameters by name:
configs/default/train.yaml:9-29 (unet.f_maps 32, kernel_dim, tree_depth, adaptive_depth).

Architecture (the reference's layer list lives in the absent wheel):
  encoder   per-point MLP on (local cell coordinate, orientation) -> trilinear splat-mean onto
            the finest level of the encoder hierarchy                        [C = f_maps channels]
  down      x_0 = relu(conv3(e_0)),  x_d = relu(conv3(mean-pool of children of x_{d-1}))
  up        top-down over the decoder hierarchy: y_d = relu(conv3(parent(y_{d+1}) + transfer(x_d)));
            a 3-way structure head (0 not-exist / 1 exist-stop / 2 exist-continue, the classes of
            models/loss.py:155-160) decides which candidate voxels exist and which are subdivided
  heads     basis = e_0 + W_b y,  normal = normalize(splat(input normals) + W_n y),  udf = W_u y
All convolutions are 3x3x3 submanifold sparse convolutions on the fp32 matrix cores
(csrc/nn.hip).  The pretrained weights are fetched from the network at run time by the reference
(models/nksr_net.py:36-38) and are unavailable offline: the default initialisation is analytic +
seeded -- trunk weights are He-random (seeded), the residual heads are zero and the structure head
is biased to "exist-continue", which makes an untrained network a sound non-learned kernel solver.
A user-supplied ``state_dict`` replaces all of it.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import call, ptr, stream
from ..svh import SparseFeatureHierarchy, SparseGrid


NORMAL_MIN_LENGTH = 1e-2     # a splatted mean normal shorter than this carries no orientation
NORMAL_MIN_WEIGHT = 1e-3     # nor does the splat of a voxel the cloud barely touches (total trilinear weight)


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
        self.trunk_features = [None] * depth


class EncodedCloud:
    """Output of ``network.encoder``: Morton-sorted cloud + the encoded finest-level features."""

    def __init__(self):
        self.xyz = self.feat = self.keys = self.voxel_feat = None


# ---- thin wrappers over csrc/nn.hip ---------------------------------------------------------------------
def sort_cloud(xyz, feat, inv_w0):
    n = xyz.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device)
    call('nksr_point_keys', ptr(xyz), n, inv_w0, ptr(keys), stream())
    ks, perm = ops.sort_pairs(keys, torch.arange(n, dtype=torch.int32, device=xyz.device), level=0)
    perm = perm.long()
    return ks, xyz[perm].contiguous(), (feat[perm].contiguous() if feat is not None else None)


def site_ranges(site_keys, grid, shift_level):
    n = grid.num_voxels
    st = torch.empty(n, dtype=torch.int32, device=grid.device)
    en = torch.empty(n, dtype=torch.int32, device=grid.device)
    call('nksr_site_ranges', ptr(site_keys), site_keys.numel(), ptr(grid.keys), n, shift_level, ptr(st), ptr(en), stream())
    return st, en


def splat_trilinear(grid, d, inv_w0, site_keys, xyz_sorted, feat_sorted):
    """Weighted SUM (and weight sum) of <= 8-channel features splatted onto level-d ``grid``."""
    n, C_ = grid.num_voxels, feat_sorted.shape[1]
    st, en = site_ranges(site_keys, grid, d)
    out = torch.empty((n, C_), dtype=torch.float32, device=grid.device)
    ws = torch.empty(n, dtype=torch.float32, device=grid.device)
    call('nksr_splat_trilinear', ptr(xyz_sorted), ptr(feat_sorted), C_, ptr(st), ptr(en), ptr(grid.nbr), ptr(grid.ijk), n,
         float(inv_w0 * 2.0 ** (-d)), ptr(out), ptr(ws), stream())
    return out, ws


def splat_mean(grid, d, inv_w0, site_keys, xyz_sorted, feat_sorted):
    n, C_ = grid.num_voxels, feat_sorted.shape[1]
    st, en = site_ranges(site_keys, grid, d)
    out = torch.empty((n, C_), dtype=torch.float32, device=grid.device)
    call('nksr_splat_mean', ptr(xyz_sorted), ptr(feat_sorted), C_, ptr(st), ptr(en), ptr(grid.nbr), ptr(grid.ijk), n,
         float(inv_w0 * 2.0 ** (-d)), ptr(out), stream())
    return out


def splat_plane(grid, d, inv_w0, site_keys, xyz_sorted, normal_sorted):
    """Plane features [n, 8] = (occupied, centroid offset, unit mean normal, 0) of level-d ``grid``."""
    n = grid.num_voxels
    st, en = site_ranges(site_keys, grid, d)
    out = torch.empty((n, 8), dtype=torch.float32, device=grid.device)
    call('nksr_splat_plane', ptr(xyz_sorted), ptr(normal_sorted), ptr(st), ptr(en), ptr(grid.nbr), ptr(grid.ijk), n,
         float(inv_w0 * 2.0 ** (-d)), ptr(out), stream())
    return out


def gather_rows(src, idx, add=None):
    n, C_ = idx.numel(), src.shape[1]
    out = torch.empty((n, C_), dtype=torch.float32, device=idx.device)
    call('nksr_gather_rows', ptr(src), ptr(idx), n, C_, ptr(add), ptr(out), stream())
    return out


class SparseConv3(nn.Module):
    """3x3x3 submanifold sparse convolution, weight [27, C_in, C_out]."""

    def __init__(self, channels, generator=None):
        super().__init__()
        self.channels = channels
        self.weight = nn.Parameter(torch.randn(27, channels, channels, generator=generator) * (2.0 / (27 * channels)) ** 0.5)
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x, nbr, relu=True, residual=None):
        n = x.shape[0]
        out = torch.empty_like(x)
        call('nksr_sparse_conv3', ptr(x.contiguous()), ptr(nbr), n, self.channels, ptr(self.weight.detach().contiguous()),
             ptr(self.bias.detach().contiguous()), ptr(residual), int(relu), ptr(out), stream())
        return out


class Head(nn.Module):
    def __init__(self, cin, cout, scale, generator=None, bias=None):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(float(scale) * torch.randn(cout, cin, generator=generator) / cin ** 0.5)
        self.bias = nn.Parameter(torch.zeros(cout) if bias is None else torch.tensor(bias, dtype=torch.float32))

    def forward(self, x):
        out = torch.empty((x.shape[0], self.cout), dtype=torch.float32, device=x.device)
        call('nksr_linear', ptr(x.contiguous()), x.shape[0], self.cin, ptr(self.weight.detach().contiguous()),
             ptr(self.bias.detach().contiguous()), self.cout, ptr(out), stream())
        return out


class PointEncoder(nn.Module):
    def __init__(self, hparams, generator=None):
        super().__init__()
        self.hparams = hparams
        C_ = int(hparams.unet.f_maps)
        self.channels = C_
        self.W1 = nn.Parameter(torch.randn(C_, 6, generator=generator) / 6 ** 0.5)
        self.b1 = nn.Parameter(torch.zeros(C_))
        self.W2 = nn.Parameter(torch.randn(C_, C_, generator=generator) / C_ ** 0.5)
        self.b2 = nn.Parameter(torch.zeros(C_))

    def forward(self, xyz, feat, svh, depth=0, sorted_keys=None):
        """``sorted_keys``: level-0 Morton keys of an ALREADY Morton-sorted cloud (skips the sort)."""
        if feat is None:
            raise RuntimeError("this network needs an orientation feature (hparams.feature='normal')")
        enc = EncodedCloud()
        if sorted_keys is not None:
            enc.keys, enc.xyz, enc.feat = sorted_keys, xyz.contiguous(), feat.to(torch.float32).contiguous()
        else:
            enc.keys, enc.xyz, enc.feat = sort_cloud(xyz.contiguous(), feat.to(torch.float32).contiguous(), svh.inv_w0)
        if self.training and torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            from .backward import EncoderFunction          # training: the voxel features carry a graph into the encoder's parameters
            enc.voxel_feat = EncoderFunction.apply(self, enc, svh, depth, self.W1, self.b1, self.W2, self.b2)
        else:
            enc.voxel_feat = self._forward_hip(enc, svh, depth)
        return enc

    def _forward_hip(self, enc, svh, depth):
        n = enc.xyz.shape[0]
        g = torch.empty((n, self.channels), dtype=torch.float32, device=enc.xyz.device)
        call('nksr_point_mlp', ptr(enc.xyz), ptr(enc.feat), n, svh.inv_w0, self.channels, ptr(self.W1.detach().contiguous()),
             ptr(self.b1.detach().contiguous()), ptr(self.W2.detach().contiguous()), ptr(self.b2.detach().contiguous()), ptr(g), stream())
        return splat_mean(svh.level(depth), depth, svh.inv_w0, enc.keys, enc.xyz, g)


class StructureUNet(nn.Module):
    def __init__(self, hparams, generator=None):
        super().__init__()
        self.hparams = hparams
        C_, D, K = int(hparams.unet.f_maps), int(hparams.tree_depth), int(hparams.kernel_dim)
        hs = float(getattr(hparams, 'head_init_scale', 0.0))
        self.down = nn.ModuleList([SparseConv3(C_, generator) for _ in range(D)])
        self.up = nn.ModuleList([SparseConv3(C_, generator) for _ in range(D)])
        self.structure_heads = nn.ModuleList([Head(C_, 3, hs, generator, bias=[-1.0, 0.0, 1.0]) for _ in range(D)])
        self.basis_heads = nn.ModuleList([Head(C_, K, hs, generator) for _ in range(D)])
        self.normal_heads = nn.ModuleList([Head(C_, 3, hs, generator) for _ in range(D)])
        self.udf_heads = nn.ModuleList([Head(C_, 8, hs, generator) for _ in range(D)])

    def forward(self, enc, enc_svh, adaptive_depth=1, gt_decoder_svh=None):
        if self.training and torch.is_grad_enabled() and (enc.voxel_feat.requires_grad or any(q.requires_grad for q in self.parameters())):
            # training (models/nksr_net.py:74-78 under autograd): same HIP forward, its outputs tied to the reverse sweep of nn/backward.py
            from .backward import UNetFunction
            tape = {}
            outs = UNetFunction.apply(self, enc, enc_svh, adaptive_depth, gt_decoder_svh, tape, enc.voxel_feat,
                                      *[q for _, q in self.named_parameters()])
            feat, dec_svh = tape['feat'], tape['dec_svh']
            for (kind, d), o in zip(tape['layout'], outs):
                if kind == 'structure':
                    feat.structure_features[d] = o
                else:
                    getattr(feat, kind + '_features')[d] = o
            return feat, dec_svh, dec_svh
        return self._forward_impl(enc, enc_svh, adaptive_depth, gt_decoder_svh, None)

    def _forward_impl(self, enc, enc_svh, adaptive_depth, gt_decoder_svh, tape):
        """``tape`` (dict or None): what the reverse sweep (nn/backward.py) needs from this forward."""
        hp = self.hparams
        D = enc_svh.depth
        dev = enc_svh.device
        K = int(hp.kernel_dim)
        vf = enc.voxel_feat.detach()
        if tape is not None:
            tape.update(x=None, pool=[vf] + [None] * (D - 1), enc_nbr=[enc_svh.level(d).nbr for d in range(D)], ranges=[None] * D,
                        dec=[None] * D, normal=[None] * D, trunk=None)
        # ---- down path on the encoder hierarchy ----------------------------------------------------
        x = [None] * D
        g = enc_svh.level(0)
        x[0] = self.down[0](vf, g.nbr)
        for d in range(1, D):
            g, gc = enc_svh.level(d), enc_svh.level(d - 1)
            st, en = site_ranges(gc.keys, g, 1)      # children = contiguous Morton range one level down
            p = torch.empty((g.num_voxels, x[d - 1].shape[1]), dtype=torch.float32, device=dev)
            call('nksr_pool_children', ptr(x[d - 1]), ptr(st), ptr(en), g.num_voxels, p.shape[1], ptr(p), stream())
            x[d] = self.down[d](p, g.nbr)
            if tape is not None:
                tape['pool'][d], tape['ranges'][d] = p, (st, en)
        if tape is not None:
            tape['x'] = x
        # ---- candidate decoder structure ------------------------------------------------------------
        cand = gt_decoder_svh if gt_decoder_svh is not None else \
            SparseFeatureHierarchy(enc_svh.voxel_size, D, dev).build_point_neighborhood_sorted(enc.keys, getattr(enc, 'cells', None))
        feat = FeatureSet(D)
        dec_levels = [None] * D
        y_up, keep_up = None, None           # trunk features / "continue" flags of the level above
        for d in range(D - 1, -1, -1):
            gc = cand.level(d)
            keys = gc.keys
            if d < D - 1:
                par = dec_levels[d + 1].hash.query((keys >> 3).contiguous())
                ok = (par >= 0) & keep_up[par.clamp_min(0).long()]
                if not bool(ok.all()):
                    keys = keys[ok].contiguous()
                    gc = SparseGrid(keys, d, enc_svh.voxel_size, coarse=dec_levels[d + 1])
                    par = dec_levels[d + 1].hash.query((keys >> 3).contiguous())
            je = enc_svh.level(d).hash.query(keys)
            t = gather_rows(x[d], je)
            if d < D - 1:
                t = gather_rows(y_up, par, add=t)
            y = self.up[d](t, gc.nbr)
            s = self.structure_heads[d](y)
            status = s.argmax(1)
            exist = status != 0
            if tape is not None:
                tape['dec'][d] = dict(je=je, par=par if d < D - 1 else None, t=t, nbr=gc.nbr, y_pre=y, exist=None if bool(exist.all()) else exist)
            if not bool(exist.all()):          # prune "not-exist" voxels (needs a re-indexed grid)
                y, s, status = y[exist].contiguous(), s[exist].contiguous(), status[exist]
                gc = SparseGrid(keys[exist].contiguous(), d, enc_svh.voxel_size, coarse=dec_levels[d + 1] if d < D - 1 else None)
            dec_levels[d] = gc
            feat.structure_features[d] = s
            feat.trunk_features[d] = y
            y_up, keep_up = y, status == 2
        dec_svh = SparseFeatureHierarchy(enc_svh.voxel_size, D, dev)
        dec_svh._levels = dec_levels
        # ---- heads -----------------------------------------------------------------------------------
        e0 = torch.zeros(K, dtype=torch.float32, device=dev)
        e0[0] = 1.0
        for d in range(D):
            y = feat.trunk_features[d]
            feat.basis_features[d] = self.basis_heads[d](y) + e0
            if d < adaptive_depth:
                if bool(hp.udf.enabled):
                    # UDF branch: analytic plane features (+ the learned head, zero at init)
                    pl = splat_plane(cand.level(d), d, enc_svh.inv_w0, enc.keys, enc.xyz, enc.feat)
                    if dec_levels[d] is not cand.level(d):
                        pl = gather_rows(pl, cand.level(d).hash.query(dec_levels[d].keys))
                    feat.udf_features[d] = pl + self.udf_heads[d](y)
                # splat on the CANDIDATE grid (it holds every cell that contains a point; the gather
                # form of the splat walks neighbour voxels), then keep the rows of surviving voxels
                s, ws = splat_trilinear(cand.level(d), d, enc_svh.inv_w0, enc.keys, enc.xyz, enc.feat)
                if dec_levels[d] is not cand.level(d):
                    sel = cand.level(d).hash.query(dec_levels[d].keys)
                    s, ws = gather_rows(s, sel), ws[sel.long()]
                nv = s + self.normal_heads[d](y)
                # unit length -- unless the splatted normals cancel (|sum w n| < 1e-2 sum w: both sides of a thin sheet in one voxel)
                # or the voxel is barely touched (sum w < 1e-3: a point on the edge of its stencil weighs 0 or 1e-8 depending on
                # rounding): that vector is noise; it stays short instead of becoming an arbitrary unit target
                nml = float(getattr(hp, 'normal_min_length', NORMAL_MIN_LENGTH))
                nmw = float(getattr(hp, 'normal_min_weight', NORMAL_MIN_WEIGHT))
                nn_ = nv.norm(dim=1)
                den = torch.maximum(nn_, nml * ws).clamp_min(max(nmw, 1e-30))      # (0 / 0: plain unit normalisation)
                feat.normal_features[d] = nv / den[:, None]
                if tape is not None:
                    tape['normal'][d] = (nv, den, nn_ >= den)
        if tape is not None:
            tape['trunk'] = feat.trunk_features
        return feat, dec_svh, dec_svh


class UDFDecoder(nn.Module):
    """``network.udf_decoder`` (models/nksr_net.py:127): decodes the unsigned distance to the input
    surface from per-voxel plane features, finest level first (csrc/nn.hip: k_udf_decode).  The
    decoder is fixed-function; what is learned lives in the U-Net's udf head."""
    FAR = 1e30

    def forward(self, xyz, svh, features):
        xyz = xyz.contiguous()
        n = xyz.shape[0]
        out = torch.full((n,), self.FAR, dtype=torch.float32, device=xyz.device)
        first = True
        for d in range(svh.depth):
            if d >= len(features) or features[d] is None:
                continue
            g = svh.level(d)
            lv = _lib.LevelT()
            lv.n, lv.offset = g.num_voxels, 0
            lv.keys, lv.ijk, lv.nbr = ptr(g.keys), ptr(g.ijk), ptr(g.nbr)
            lv.hkeys, lv.hvals, lv.hcap = ptr(g.hash.hkeys), ptr(g.hash.hvals), g.hash.cap
            f = features[d].contiguous()
            call('nksr_udf_decode', C.byref(lv), d, ptr(f), ptr(xyz), n, float(svh.inv_w0 * 2.0 ** (-d)),
                 float(svh.voxel_size * (1 << d)), 0 if first else 1, ptr(out), stream())
            first = False
        return out


class NKSRNetwork(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        gen = torch.Generator().manual_seed(int(getattr(hparams, 'seed', 0)))
        self.interpolators = nn.ModuleList([
            Interpolator(hparams.kernel_dim, hparams.interpolator.hidden_dim, hparams.interpolator.n_hidden,
                         init_scale=float(getattr(hparams, 'interpolator_init_scale', 0.0)), generator=gen)
            for _ in range(hparams.tree_depth)])
        self.encoder = PointEncoder(hparams, gen)
        self.unet = StructureUNet(hparams, gen)
        self.sdf_decoder = None
        self.udf_decoder = UDFDecoder() if bool(hparams.udf.enabled) else None
