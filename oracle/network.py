"""Oracle: the sparse feature-hierarchy network (restates nksr_amd.nn.network, which mirrors
``nksr.NKSRNetwork``: encoder / unet call sites models/nksr_net.py:73-78, hyper-parameters
configs/default/train.yaml:9-29).  numpy only; weights are passed in as a dict of arrays exported
from the torch module's ``state_dict`` so both sides run the same parameters."""
import numpy as np

from . import hierarchy, spec


def export_params(net):
    """torch NKSRNetwork -> {name: float32 ndarray}."""
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in net.state_dict().items()}


def conv3(x, nbr, W, b, relu=True):
    out = np.tile(b.astype(np.float32), (x.shape[0], 1))
    xp = np.concatenate([x, np.zeros((1, x.shape[1]), np.float32)])   # index -1 -> zero row
    for s in range(27):
        out += xp[nbr[:, s]] @ W[s]
    return np.maximum(out, 0) if relu else out


NORMAL_MIN_LENGTH = 1e-2     # a splatted mean normal shorter than this carries no orientation: not blown up to unit length
NORMAL_MIN_WEIGHT = 1e-3     # nor is the splat of a voxel the cloud barely touches (total trilinear weight below this)


def splat(level, xyz, feat, voxel_size_d, mean, return_weights=False):
    inv_w = np.float32(spec.inv_w0_f32(voxel_size_d))
    p = xyz.astype(np.float32) * inv_w
    base = np.floor(p - np.float32(0.5)).astype(np.int32)
    acc = np.zeros((level.n, feat.shape[1]), np.float64)
    ws = np.zeros(level.n, np.float64)
    for co in spec.CORNER_OFFSETS:
        ijk = base + co[None]
        w = np.prod(np.float32(1.0) - np.abs(p - (ijk.astype(np.float32) + np.float32(0.5))), axis=1)
        j = level.lookup(ijk)
        ok = (j >= 0) & (w > 0)
        np.add.at(acc, j[ok], feat[ok].astype(np.float64) * w[ok, None])
        np.add.at(ws, j[ok], w[ok])
    if mean:
        acc = acc / np.where(ws > 0, ws, 1.0)[:, None] * (ws > 0)[:, None]
    if return_weights:
        return acc.astype(np.float32), ws.astype(np.float32)
    return acc.astype(np.float32)


def forward(P, xyz, normal, voxel_size, depth, kernel_dim, adaptive_depth, udf=False):
    """Returns (dec hierarchy, basis_features, normal_features, structure logits, trunk features)."""
    enc = hierarchy.Hierarchy(voxel_size, depth).build_point_splatting(xyz)
    cand = hierarchy.Hierarchy(voxel_size, depth).build_point_neighborhood(xyz)
    # encoder
    H0, p = spec.half_index(xyz, voxel_size)
    u = (p - (H0 >> 1).astype(np.float32)).astype(np.float32)
    inp = np.concatenate([u - np.float32(0.5), normal.astype(np.float32)], 1)
    h = np.maximum(inp @ P['encoder.W1'].T + P['encoder.b1'], 0)
    g = (h @ P['encoder.W2'].T + P['encoder.b2']).astype(np.float32)
    e0 = splat(enc.levels[0], xyz, g, voxel_size, mean=True)
    # down path
    x = [conv3(e0, enc.levels[0].nbr, P['unet.down.0.weight'], P['unet.down.0.bias'])]
    for d in range(1, depth):
        Lc, Lp = enc.levels[d - 1], enc.levels[d]
        par = Lp.lookup(Lc.ijk >> 1)
        acc = np.zeros((Lp.n, e0.shape[1]), np.float64)
        cnt = np.zeros(Lp.n)
        ok = par >= 0
        np.add.at(acc, par[ok], x[d - 1][ok].astype(np.float64))
        np.add.at(cnt, par[ok], 1)
        pool = (acc / np.maximum(cnt, 1)[:, None]).astype(np.float32)
        x.append(conv3(pool, Lp.nbr, P['unet.down.%d.weight' % d], P['unet.down.%d.bias' % d]))
    # top-down decoder
    levels, trunk, logits = [None] * depth, [None] * depth, [None] * depth
    y_up = keep_up = None
    for d in range(depth - 1, -1, -1):
        keys = cand.levels[d].keys
        L = hierarchy.Level(keys, d, voxel_size)
        if d < depth - 1:
            par = levels[d + 1].lookup(L.ijk >> 1)
            ok = (par >= 0) & keep_up[np.maximum(par, 0)]
            L = hierarchy.Level(keys[ok], d, voxel_size)
            par = levels[d + 1].lookup(L.ijk >> 1)
        L.build_nbr()
        je = enc.levels[d].lookup(L.ijk)
        t = np.where((je >= 0)[:, None], x[d][np.maximum(je, 0)], np.float32(0)).astype(np.float32)
        if d < depth - 1:
            t = t + y_up[par]
        y = conv3(t, L.nbr, P['unet.up.%d.weight' % d], P['unet.up.%d.bias' % d])
        s = (y @ P['unet.structure_heads.%d.weight' % d].T + P['unet.structure_heads.%d.bias' % d]).astype(np.float32)
        status = s.argmax(1)
        exist = status != 0
        if not exist.all():
            y, s, status = y[exist], s[exist], status[exist]
            L = hierarchy.Level(L.keys[exist], d, voxel_size)
            L.build_nbr()
        levels[d], trunk[d], logits[d] = L, y, s
        y_up, keep_up = y, status == 2
    dec = hierarchy.Hierarchy(voxel_size, depth)
    dec.levels = levels
    dec.finalize()
    basis, normals, normal_norm = [], [None] * depth, [None] * depth
    e0v = np.zeros(kernel_dim, np.float32)
    e0v[0] = 1
    for d in range(depth):
        y = trunk[d]
        basis.append((y @ P['unet.basis_heads.%d.weight' % d].T + P['unet.basis_heads.%d.bias' % d] + e0v).astype(np.float32))
        if d < adaptive_depth:
            sN, wN = splat(levels[d], xyz, normal, voxel_size * (1 << d), mean=False, return_weights=True)
            nv = sN + (y @ P['unet.normal_heads.%d.weight' % d].T + P['unet.normal_heads.%d.bias' % d]).astype(np.float32)
            normal_norm[d] = np.linalg.norm(nv, axis=1)
            # unit length -- unless the splatted normals cancel (|sum w n| < 1e-2 sum w: opposite sides of a thin sheet in one
            # voxel) or the voxel is barely touched (sum w < 1e-3: a point ON the edge of its stencil has weight 0 or 1e-8 depending
            # on rounding): such a vector is noise and stays short instead of becoming an arbitrary unit target
            den = np.maximum(np.maximum(normal_norm[d], np.float32(NORMAL_MIN_LENGTH) * wN), np.float32(NORMAL_MIN_WEIGHT))
            normals[d] = (nv / den[:, None]).astype(np.float32)
    forward.last_udf = None
    if udf:     # UDF branch: plane features + the (zero-initialised) learned head
        forward.last_udf = [None] * depth
        for d in range(min(adaptive_depth, depth)):
            head = (trunk[d] @ P['unet.udf_heads.%d.weight' % d].T + P['unet.udf_heads.%d.bias' % d]).astype(np.float32)
            forward.last_udf[d] = (plane_features(levels[d], xyz, normal, voxel_size * (1 << d)) + head).astype(np.float32)
    forward.last_normal_norm = normal_norm   # conditioning of the normalisation (used by the parity test)
    return dec, basis, normals, logits, trunk


# ---- UDF mask branch (NeuralField + network.udf_decoder, models/nksr_net.py:124-130) -------------------
UDF_FAR = np.float32(1e30)


def plane_features(level, xyz, normal, voxel_size_d):
    """[n, 8] = (occupied, trilinear-weighted centroid offset in voxel units, unit mean normal, 0)."""
    inv_w = np.float32(spec.inv_w0_f32(voxel_size_d))
    p = xyz.astype(np.float32) * inv_w
    base = np.floor(p - np.float32(0.5)).astype(np.int32)
    acc = np.zeros((level.n, 6), np.float64)
    ws = np.zeros(level.n, np.float64)
    for co in spec.CORNER_OFFSETS:
        ijk = base + co[None]
        r = (p - (ijk.astype(np.float32) + np.float32(0.5))).astype(np.float32)
        w = np.prod(np.float32(1.0) - np.abs(r), axis=1)
        j = level.lookup(ijk)
        ok = (j >= 0) & (w > 0)
        np.add.at(acc, j[ok], np.concatenate([r[ok], normal[ok].astype(np.float32)], 1).astype(np.float64) * w[ok, None])
        np.add.at(ws, j[ok], w[ok])
    out = np.zeros((level.n, 8), np.float32)
    occ = ws > 0
    out[:, 0] = occ
    out[:, 1:4] = (acc[:, :3] / np.where(occ, ws, 1.0)[:, None]) * occ[:, None]
    nn_ = np.linalg.norm(acc[:, 3:], axis=1)
    out[:, 4:7] = acc[:, 3:] / np.where(nn_ > 1e-8, nn_, 1.0)[:, None] * (nn_ > 1e-8)[:, None]
    return out


def udf_decode(hier, feats, xyz):
    """Unsigned plane distance, finest level with an occupied surrounding voxel first."""
    out = np.full(xyz.shape[0], UDF_FAR, np.float32)
    for d, L in enumerate(hier.levels):
        if d >= len(feats) or feats[d] is None:
            continue
        w_d = hier.voxel_size * (1 << d)
        inv_w = np.float32(spec.inv_w0_f32(hier.voxel_size) * np.float32(2.0 ** (-d)))
        p = xyz.astype(np.float32) * inv_w
        fl = np.floor(p - np.float32(0.5))
        base = fl.astype(np.int32)
        v = (p - np.float32(0.5) - fl).astype(np.float32)
        sw = np.zeros(xyz.shape[0], np.float32)
        sd = np.zeros(xyz.shape[0], np.float32)
        for co in spec.CORNER_OFFSETS:
            ijk = base + co[None]
            j = L.lookup(ijk)
            f = feats[d][np.maximum(j, 0)]
            ok = (j >= 0) & (f[:, 0] > 0.5)
            t = np.prod(np.where(co[None] == 1, v, np.float32(1.0) - v), axis=1).astype(np.float32)
            r = (p - (ijk.astype(np.float32) + np.float32(0.5)) - f[:, 1:4]).astype(np.float32)
            dist = (r * f[:, 4:7]).sum(1).astype(np.float32)
            sw += np.where(ok, t, np.float32(0))
            sd += np.where(ok, t * dist, np.float32(0))
        val = (np.abs(sd / np.where(sw > 0, sw, np.float32(1))) * np.float32(w_d)).astype(np.float32)
        take = (out >= np.float32(0.5) * UDF_FAR) & (sw > 0)
        out[take] = val[take]
    return out
