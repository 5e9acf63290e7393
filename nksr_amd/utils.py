"""Helpers the reference's examples lean on third-party packages for:
binary/ascii PLY reader + writer (pycg.vis.from_file / to_file, examples/common.py:21-73),
free-memory warning (examples/common.py:77-88) and deterministic synthetic clouds standing
in for the downloadable assets (no network here)."""
import numpy as np
import torch

_PLY_TYPES = {'char': 'i1', 'uchar': 'u1', 'short': 'i2', 'ushort': 'u2', 'int': 'i4', 'uint': 'u4', 'float': 'f4',
              'double': 'f8', 'int8': 'i1', 'uint8': 'u1', 'int16': 'i2', 'uint16': 'u2', 'int32': 'i4',
              'uint32': 'u4', 'float32': 'f4', 'float64': 'f8'}


def read_ply(path):
    """Returns a dict of per-vertex property arrays (x, y, z, nx, ..., red, ..., sensor_x, ...)."""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise RuntimeError('%s is not a PLY file' % path)
        fmt, props, nvert, in_vertex = None, [], 0, False
        while True:
            line = f.readline()
            if not line:
                raise RuntimeError('unterminated PLY header')
            tok = line.decode('ascii', 'replace').split()
            if not tok:
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    nvert = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise RuntimeError('list property on vertices is not supported')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        if fmt == 'ascii':
            data = np.loadtxt(f, max_rows=nvert, ndmin=2)
            return {name: data[:, i].astype(t) for i, (name, t) in enumerate(props)}
        end = '<' if fmt == 'binary_little_endian' else '>'
        dt = np.dtype([(n, end + t) for n, t in props])
        arr = np.frombuffer(f.read(dt.itemsize * nvert), dtype=dt, count=nvert)
        return {n: np.ascontiguousarray(arr[n]) for n, _ in props}


def load_point_cloud(path):
    """(xyz [N,3] f32, normal [N,3] f32 or None, color [N,3] f32 in [0,1] or None, sensor or None)."""
    p = read_ply(path)
    xyz = np.stack([p['x'], p['y'], p['z']], 1).astype(np.float32)
    nrm = np.stack([p['nx'], p['ny'], p['nz']], 1).astype(np.float32) if 'nx' in p else None
    col = np.stack([p['red'], p['green'], p['blue']], 1).astype(np.float32) / 255.0 if 'red' in p else None
    sen = np.stack([p['sensor_x'], p['sensor_y'], p['sensor_z']], 1).astype(np.float32) if 'sensor_x' in p else None
    return xyz, nrm, col, sen


def write_ply_mesh(path, v, f, c=None):
    v = np.asarray(v.detach().cpu() if torch.is_tensor(v) else v, np.float32)
    f = np.asarray(f.detach().cpu() if torch.is_tensor(f) else f, np.int32)
    with open(path, 'wb') as out:
        hdr = ['ply', 'format binary_little_endian 1.0', 'element vertex %d' % len(v), 'property float x',
               'property float y', 'property float z']
        if c is not None:
            hdr += ['property uchar red', 'property uchar green', 'property uchar blue']
        hdr += ['element face %d' % len(f), 'property list uchar int vertex_indices', 'end_header']
        out.write(('\n'.join(hdr) + '\n').encode())
        if c is not None:
            c8 = (np.clip(np.asarray(c.detach().cpu() if torch.is_tensor(c) else c), 0, 1) * 255).astype(np.uint8)
            rec = np.empty(len(v), dtype=[('p', '<f4', 3), ('c', 'u1', 3)])
            rec['p'], rec['c'] = v, c8
            out.write(rec.tobytes())
        else:
            out.write(v.astype('<f4').tobytes())
        rec = np.empty(len(f), dtype=[('n', 'u1'), ('i', '<i4', 3)])
        rec['n'], rec['i'] = 3, f
        out.write(rec.tobytes())


def write_obj_mesh(path, v, f):
    """Wavefront OBJ (what the reference's examples write through pycg.vis.to_file, examples/gis_app.py:55).  ``v`` may be float64 in
    a projected coordinate system (1e5 .. 1e6 metres): positions are written with 6 decimals, not through fp32."""
    v = np.asarray(v.detach().cpu() if torch.is_tensor(v) else v, np.float64)
    f = np.asarray(f.detach().cpu() if torch.is_tensor(f) else f, np.int64) + 1
    with open(path, 'w') as out:
        out.write(''.join('v %.6f %.6f %.6f\n' % (p[0], p[1], p[2]) for p in v))
        out.write(''.join('f %d %d %d\n' % (t[0], t[1], t[2]) for t in f))


def warning_on_low_memory(threshold_mb):
    if torch.cuda.is_available():
        free, _ = torch.cuda.mem_get_info()
        if free / 2 ** 20 < threshold_mb:
            print('[nksr_amd] warning: only %.0f MB of free GPU memory (< %.0f MB)' % (free / 2 ** 20, threshold_mb))


# ---- deterministic synthetic clouds (stand-ins for the reference's downloadable assets) ----------
def synth_sphere(n, radius=0.45, noise=0.0, seed=0, center=(0.0, 0.0, 0.0)):
    rs = np.random.RandomState(seed)
    d = rs.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = d * radius + np.asarray(center)
    if noise > 0:
        xyz = xyz + rs.randn(n, 3) * noise
    return xyz.astype(np.float32), d.astype(np.float32)


def synth_torus(n, R=0.32, r=0.12, noise=0.0, seed=0, center=(0.0, 0.0, 0.0)):
    rs = np.random.RandomState(seed)
    # rejection-sample for area-uniform density
    u = rs.rand(3 * n) * 2 * np.pi
    v = rs.rand(3 * n) * 2 * np.pi
    keep = rs.rand(3 * n) < (R + r * np.cos(v)) / (R + r)
    u, v = u[keep][:n], v[keep][:n]
    nrm = np.stack([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)], 1)
    xyz = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], 1) + np.asarray(center)
    if noise > 0:
        xyz = xyz + rs.randn(len(xyz), 3) * noise
    return xyz.astype(np.float32), nrm.astype(np.float32)


def synth_scene(n, seed=0, extent=(40.0, 40.0, 10.0), noise=0.01, n_objects=8, origin=(0.0, 0.0, 0.0)):
    """oriented points on a union of spheres / tori laid out in
    a box, analytic normals, Gaussian position noise."""
    rs = np.random.RandomState(seed)
    per = [n // n_objects + (1 if i < n % n_objects else 0) for i in range(n_objects)]
    pts, nrms = [], []
    gx = int(np.ceil(np.sqrt(n_objects)))
    for i, m in enumerate(per):
        cx = (i % gx + 0.5) * extent[0] / gx
        cy = (i // gx + 0.5) * extent[1] / gx
        cz = extent[2] * 0.5
        size = min(extent[0] / gx, extent[1] / gx, extent[2]) * 0.45
        if i % 2 == 0:
            p, q = synth_sphere(m, radius=size, seed=seed * 1000 + i)
        else:
            p, q = synth_torus(m, R=size * 0.7, r=size * 0.28, seed=seed * 1000 + i)
        pts.append(p + np.array([cx, cy, cz], np.float32))
        nrms.append(q)
    xyz = np.concatenate(pts) + rs.randn(n, 3).astype(np.float32) * noise + np.asarray(origin, np.float32)
    return xyz.astype(np.float32), np.concatenate(nrms).astype(np.float32)


def synth_terrain(n, seed=0, extent=(1000.0, 1000.0), origin=(0.0, 0.0), amp=8.0):
    """km-scale height field (sum of sinusoids), analytic normals."""
    rs = np.random.RandomState(seed)
    x = rs.rand(n) * extent[0] + origin[0]
    y = rs.rand(n) * extent[1] + origin[1]
    fr = np.random.RandomState(12345).rand(6, 3)  # scene-global frequencies: chunks agree
    z = np.zeros(n)
    dzdx = np.zeros(n)
    dzdy = np.zeros(n)
    for k in range(6):
        kx, ky, ph = (fr[k, 0] + 0.2) * 0.05, (fr[k, 1] + 0.2) * 0.05, fr[k, 2] * 6.28
        a = amp / (k + 1)
        z += a * np.sin(kx * x + ky * y + ph)
        c = a * np.cos(kx * x + ky * y + ph)
        dzdx += c * kx
        dzdy += c * ky
    nrm = np.stack([-dzdx, -dzdy, np.ones(n)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.stack([x, y, z], 1).astype(np.float32), nrm.astype(np.float32)


def synth_street(n, seed=0, extent=(200.0, 100.0), n_boxes=12, n_poles=10, noise=0.0, sensor_height=1.8):
    """SURVEY.md section 8d config 4 (CARLA stand-in): a street scene -- gently undulating ground, box
    buildings on both sides of a centre line along x, vertical cylinders (poles) -- sampled area-uniformly,
    with the scanner positions on the centre line (the point's x, clamped).  Returns
    (xyz [n,3], analytic normal [n,3], sensor [n,3]); sensor-only pipelines ignore the normals."""
    rs = np.random.RandomState(seed)
    ex, ey = float(extent[0]), float(extent[1])
    s = min(ex, ey)
    boxes = []
    for i in range(n_boxes):
        side = 1.0 if i % 2 == 0 else -1.0
        w, d, h = s * (0.10 + 0.06 * rs.rand()), s * (0.10 + 0.05 * rs.rand()), s * (0.08 + 0.10 * rs.rand())
        cx = ex * (i // 2 + 0.5 + 0.2 * (rs.rand() - 0.5)) / max((n_boxes + 1) // 2, 1)
        cy = ey * 0.5 + side * (ey * 0.17 + d * 0.5 + ey * 0.08 * rs.rand())
        boxes.append((cx - w / 2, cx + w / 2, max(cy - d / 2, 0.02 * ey), min(cy + d / 2, 0.98 * ey), h))
    poles = [(ex * (j + 0.5) / n_poles, ey * 0.5 + (1.0 if j % 2 else -1.0) * ey * 0.10, s * 0.012 + 0.03, s * 0.05 + 1.0)
             for j in range(n_poles)]
    area = [ex * ey] + [2 * (b[1] - b[0]) * b[4] + 2 * (b[3] - b[2]) * b[4] + (b[1] - b[0]) * (b[3] - b[2]) for b in boxes] + \
           [2 * np.pi * p[2] * p[3] for p in poles]
    cnt = np.floor(np.asarray(area) / sum(area) * n).astype(int)
    cnt[0] += n - cnt.sum()
    pts, nrms = [], []
    # ground (points under a building are not visible)
    m = int(cnt[0] * 1.5) + 64
    gx, gy = rs.rand(m) * ex, rs.rand(m) * ey
    hidden = np.zeros(m, bool)
    for b in boxes:
        hidden |= (gx > b[0]) & (gx < b[1]) & (gy > b[2]) & (gy < b[3])
    gx, gy = gx[~hidden][:cnt[0]], gy[~hidden][:cnt[0]]
    cnt[0] = len(gx)
    ka, kb, amp = 2 * np.pi / (0.35 * s), 2 * np.pi / (0.5 * s), 0.012 * s
    gz = amp * np.sin(ka * gx) * np.cos(kb * gy)
    gn = np.stack([-amp * ka * np.cos(ka * gx) * np.cos(kb * gy), amp * kb * np.sin(ka * gx) * np.sin(kb * gy), np.ones_like(gx)], 1)
    pts.append(np.stack([gx, gy, gz], 1))
    nrms.append(gn / np.linalg.norm(gn, axis=1, keepdims=True))
    for b, c in zip(boxes, cnt[1:1 + len(boxes)]):
        w, d, h = b[1] - b[0], b[3] - b[2], b[4]
        fa = np.array([w * h, w * h, d * h, d * h, w * d])
        face = rs.choice(5, size=c, p=fa / fa.sum())
        u, v = rs.rand(c), rs.rand(c)
        p = np.zeros((c, 3))
        q = np.zeros((c, 3))
        for f_, (ax, val, sg) in enumerate([(1, b[2], -1), (1, b[3], 1), (0, b[0], -1), (0, b[1], 1)]):
            k = face == f_
            o = 1 - ax
            p[k, ax] = val
            p[k, o] = (b[0] + u[k] * w) if o == 0 else (b[2] + u[k] * d)
            p[k, 2] = v[k] * h
            q[k, ax] = sg
        k = face == 4
        p[k, 0], p[k, 1], p[k, 2] = b[0] + u[k] * w, b[2] + v[k] * d, h
        q[k, 2] = 1
        pts.append(p)
        nrms.append(q)
    for pl, c in zip(poles, cnt[1 + len(boxes):]):
        th, z = rs.rand(c) * 2 * np.pi, rs.rand(c) * pl[3]
        q = np.stack([np.cos(th), np.sin(th), np.zeros(c)], 1)
        pts.append(np.stack([pl[0] + pl[2] * q[:, 0], pl[1] + pl[2] * q[:, 1], z], 1))
        nrms.append(q)
    xyz = np.concatenate(pts)
    nrm = np.concatenate(nrms)
    if noise > 0:
        xyz = xyz + rs.randn(*xyz.shape) * noise
    sensor = np.stack([np.clip(xyz[:, 0], 0.0, ex), np.full(len(xyz), ey * 0.5), np.full(len(xyz), sensor_height)], 1)
    perm = rs.permutation(len(xyz))
    return xyz[perm].astype(np.float32), nrm[perm].astype(np.float32), sensor[perm].astype(np.float32)


def terrain_height(x, y, amp=8.0):
    """Scene-global height field of synth_terrain / terrain_tile (+ its gradient)."""
    fr = np.random.RandomState(12345).rand(6, 3)
    z = np.zeros_like(x, dtype=np.float64)
    dzdx = np.zeros_like(z)
    dzdy = np.zeros_like(z)
    for k in range(6):
        kx, ky, ph = (fr[k, 0] + 0.2) * 0.05, (fr[k, 1] + 0.2) * 0.05, fr[k, 2] * 6.28
        a = amp / (k + 1)
        z += a * np.sin(kx * x + ky * y + ph)
        c = a * np.cos(kx * x + ky * y + ph)
        dzdx += c * kx
        dzdy += c * ky
    return z, dzdx, dzdy


def terrain_tile(tile_xy, n, tile=125.0, seed=0, boxes_per_tile=8, amp=8.0):
    """BASELINE.json configs[4]: one ``tile`` x ``tile`` metre tile of the km-scale
    height field (+ ``boxes_per_tile`` box buildings standing on it: 8 x 64 tiles ~ the 500 boxes of the survey),
    ``n`` points, seeded by the tile index -- so a rank can generate exactly the tiles it needs and every rank
    sees the same points in a shared tile.  Returns (xyz, normal)."""
    tx, ty = int(tile_xy[0]), int(tile_xy[1])
    rs = np.random.RandomState((seed * 1_000_003 + tx * 1009 + ty) % (2 ** 31 - 1))
    ox, oy = tx * tile, ty * tile
    boxes = []
    for _ in range(boxes_per_tile):
        w, d, h = 6 + 8 * rs.rand(), 6 + 8 * rs.rand(), 4 + 10 * rs.rand()
        cx, cy = ox + w + (tile - 2 * w) * rs.rand(), oy + d + (tile - 2 * d) * rs.rand()
        z0 = float(terrain_height(np.array([cx]), np.array([cy]), amp)[0][0]) - 1.0
        boxes.append((cx - w / 2, cx + w / 2, cy - d / 2, cy + d / 2, z0, z0 + h + 1.0))
    barea = [2 * (b[1] - b[0] + b[3] - b[2]) * (b[5] - b[4]) + (b[1] - b[0]) * (b[3] - b[2]) for b in boxes]
    frac = sum(barea) / (tile * tile + sum(barea)) if boxes else 0.0
    nb = int(n * frac)
    pts, nrms = [], []
    for i, b in enumerate(boxes):
        c = int(round(nb * barea[i] / sum(barea)))
        w, d, h = b[1] - b[0], b[3] - b[2], b[5] - b[4]
        fa = np.array([w * h, w * h, d * h, d * h, w * d])
        face = rs.choice(5, size=c, p=fa / fa.sum())
        u, v = rs.rand(c), rs.rand(c)
        p, q = np.zeros((c, 3)), np.zeros((c, 3))
        for f_, (ax, val, sg) in enumerate([(1, b[2], -1), (1, b[3], 1), (0, b[0], -1), (0, b[1], 1)]):
            k = face == f_
            o = 1 - ax
            p[k, ax] = val
            p[k, o] = (b[0] + u[k] * w) if o == 0 else (b[2] + u[k] * d)
            p[k, 2] = b[4] + v[k] * h
            q[k, ax] = sg
        k = face == 4
        p[k, 0], p[k, 1], p[k, 2] = b[0] + u[k] * w, b[2] + v[k] * d, b[5]
        q[k, 2] = 1
        vis = p[:, 2] >= terrain_height(p[:, 0], p[:, 1], amp)[0]       # wall points below the terrain are not visible
        pts.append(p[vis])
        nrms.append(q[vis])
    ng = n - sum(len(p) for p in pts)                                   # the ground takes the rest: exactly n points per tile
    m = int(ng * 1.25) + 256
    x, y = ox + rs.rand(m) * tile, oy + rs.rand(m) * tile
    hid = np.zeros(m, bool)
    for b in boxes:
        hid |= (x > b[0]) & (x < b[1]) & (y > b[2]) & (y < b[3])
    x, y = x[~hid][:ng], y[~hid][:ng]
    assert len(x) == ng
    z, dzdx, dzdy = terrain_height(x, y, amp)
    gn = np.stack([-dzdx, -dzdy, np.ones_like(z)], 1)
    pts.append(np.stack([x, y, z], 1))
    nrms.append(gn / np.linalg.norm(gn, axis=1, keepdims=True))
    return np.concatenate(pts).astype(np.float32), np.concatenate(nrms).astype(np.float32)


def synth_terrain_patch(n, seed=0, extent=(14.0, 14.0), box=True):
    """Small stand-in of the configs[4] scene for oracle-sized parity cases: a wavy height field over
    ``extent`` plus one box standing on it.  Returns (xyz, normal)."""
    rs = np.random.RandomState(seed)
    ex, ey = float(extent[0]), float(extent[1])
    bx = (0.55 * ex, 0.78 * ex, 0.30 * ey, 0.52 * ey, 1.9) if box else None
    barea = (2 * (bx[1] - bx[0] + bx[3] - bx[2]) * bx[4] + (bx[1] - bx[0]) * (bx[3] - bx[2])) if box else 0.0
    nb = int(n * barea / (ex * ey + barea))
    m = int((n - nb) * 1.3) + 64
    x, y = rs.rand(m) * ex, rs.rand(m) * ey
    if box:
        hid = (x > bx[0]) & (x < bx[1]) & (y > bx[2]) & (y < bx[3])
        x, y = x[~hid], y[~hid]
    x, y = x[:n - nb], y[:n - nb]
    z = 0.6 * np.sin(0.5 * x + 0.3 * y) + 0.3 * np.cos(0.8 * y)
    gn = np.stack([-0.3 * np.cos(0.5 * x + 0.3 * y), -(0.18 * np.cos(0.5 * x + 0.3 * y) - 0.24 * np.sin(0.8 * y)), np.ones_like(x)], 1)
    pts, nrms = [np.stack([x, y, z], 1)], [gn / np.linalg.norm(gn, axis=1, keepdims=True)]
    if box:
        w, d, h = bx[1] - bx[0], bx[3] - bx[2], bx[4]
        z0 = -1.2
        fa = np.array([w * (h - z0), w * (h - z0), d * (h - z0), d * (h - z0), w * d])
        face = rs.choice(5, size=nb, p=fa / fa.sum())
        u, v = rs.rand(nb), rs.rand(nb)
        p, q = np.zeros((nb, 3)), np.zeros((nb, 3))
        for f_, (ax, val, sg) in enumerate([(1, bx[2], -1), (1, bx[3], 1), (0, bx[0], -1), (0, bx[1], 1)]):
            k = face == f_
            o = 1 - ax
            p[k, ax] = val
            p[k, o] = (bx[0] + u[k] * w) if o == 0 else (bx[2] + u[k] * d)
            p[k, 2] = z0 + v[k] * (h - z0)
            q[k, ax] = sg
        k = face == 4
        p[k, 0], p[k, 1], p[k, 2] = bx[0] + u[k] * w, bx[2] + v[k] * d, h
        q[k, 2] = 1
        vis = p[:, 2] >= 0.6 * np.sin(0.5 * p[:, 0] + 0.3 * p[:, 1]) + 0.3 * np.cos(0.8 * p[:, 1])
        pts.append(p[vis])
        nrms.append(q[vis])
    xyz, nrm = np.concatenate(pts), np.concatenate(nrms)
    perm = rs.permutation(len(xyz))
    return xyz[perm].astype(np.float32), nrm[perm].astype(np.float32)


def synth_rounded_box(n, half=(0.30, 0.22, 0.16), radius=0.10, noise=0.0, seed=0):
    """Points + analytic normals on the surface {x : dist(x, box(half)) = radius}: uniform samples of a thin
    exterior shell projected onto the surface (exact closest-point projection)."""
    rs = np.random.RandomState(seed)
    b = np.asarray(half, np.float64)
    out_p, out_n, have = [], [], 0
    while have < n:
        x = (rs.rand(4 * n, 3) * 2 - 1) * (b + radius * 1.1)
        q = np.clip(x, -b, b)
        d = np.linalg.norm(x - q, axis=1)
        k = (d > radius * 0.9) & (d < radius * 1.1)
        nn_ = (x[k] - q[k]) / d[k, None]
        out_p.append(q[k] + radius * nn_)
        out_n.append(nn_)
        have += int(k.sum())
    p, nrm = np.concatenate(out_p)[:n], np.concatenate(out_n)[:n]
    if noise > 0:
        p = p + rs.randn(n, 3) * noise
    return p.astype(np.float32), nrm.astype(np.float32)
