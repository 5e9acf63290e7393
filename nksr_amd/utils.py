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
