"""TEST INFRASTRUCTURE (CPU oracle; never imported by the product).

One FULL-SIZE chunk of the scene bench.py times (BASELINE.json configs[4]: 10 M points, 8 x 8 tiles of 125 m, tree_depth 5,
chunk_size = one tile), solved by the oracle -- the reference solves its chunks one after the other
(examples/recons_by_chunk.py:26-30), so one chunk of the batch is a complete reconstruction the numpy restatement can do:

    python -m oracle.make_golden_scene_chunk [chunk id, default 27]        (tens of minutes of CPU, ~25 GB; run in the dev container)

Writes tests/golden/scene_chunk<id>_golden.npz: the chunk's points count, per-level voxel counts and key digests, the converged
coefficients alpha (fp32, solved to 3e-7), the iteration count of the oracle's Jacobi-PCG and field values + gradients at 4 500
probes (input points and the same points half a voxel along +- their normals, in the chunk's slot of the exploded frame).  tests/test_gpu_full_size.py compares the segment of the 64-chunk batch with it (check_alpha, 1e-4 of max|alpha|:
the SURVEY.md section 8c contract) -- the bench-scale SOLUTION pinned on the oracle, not only sampled operator rows.
The inputs are regenerated from seeds (nksr_amd.utils.terrain_tile: numpy RandomState, bit-stable), not stored.
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')

TILE, TILES, N_SCENE = 125.0, 8, 10_000_000


def chunk_inputs(c, voxel_size, depth):
    """(xs, normals, scale, frame shift): the points chunk ``c`` of the scene solves, translated into its slot of the exploded
    frame -- the oracle's restatement of nksr_amd/chunking.py on bench.py's scene (terrain_setup), one chunk only."""
    from nksr_amd import utils
    from oracle import chunking, density
    per = N_SCENE // (TILES * TILES)
    t00 = utils.terrain_tile((0, 0), per, TILE, seed=0)[0]
    scale = density.scale_for_detail_level(t00, 1.0, voxel_size)
    cx, cy = c // TILES, c % TILES
    xs, ns = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if 0 <= cx + dx < TILES and 0 <= cy + dy < TILES:
                p, q = utils.terrain_tile((cx + dx, cy + dy), per, TILE, seed=0)
                xs.append(p)
                ns.append(q)
    xyz = (np.concatenate(xs) * np.float32(scale)).astype(np.float32)
    nrm = np.concatenate(ns).astype(np.float32)
    lo = [0.0, 0.0, -40.0 * scale]
    chunk_size = TILE * scale
    grid = [TILES, TILES, 1]
    wc = voxel_size * 2 ** (depth - 1)
    ov = max(0.05 * chunk_size, chunking.OV_FLOOR * wc)
    band = ov + chunking.BAND_EXTRA * wc
    frame = chunking.Frame(voxel_size, depth, lo, grid, chunk_size, band)
    clo = [lo[0] + cx * chunk_size, lo[1] + cy * chunk_size]
    m = np.ones(xyz.shape[0], bool)
    for a in range(2):
        m &= (xyz[:, a] >= np.float32(clo[a] - band)) & (xyz[:, a] < np.float32(clo[a] + chunk_size + band))
    sh = frame.shift(c)
    return (np.ascontiguousarray(xyz[m]) + sh[None]).astype(np.float32), np.ascontiguousarray(nrm[m]), scale, sh


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(c=27):
    import torch  # noqa: F401  (seeded network parameters come from the product's module definition)
    from nksr_amd import configs
    from nksr_amd.nn.network import NKSRNetwork
    from oracle import network as onet, pipeline
    hp = configs.get_hparams('ks', tree_depth=5)
    P = onet.export_params(NKSRNetwork(hp))
    t0 = time.perf_counter()
    xs, nrm, scale, sh = chunk_inputs(c, hp.voxel_size, hp.tree_depth)
    print('chunk %d: %d points (band included), scale %.9g, %.1f s' % (c, xs.shape[0], scale, time.perf_counter() - t0), flush=True)
    tm = {}
    fld = pipeline.reconstruct(xs, nrm, voxel_size=hp.voxel_size, depth=hp.tree_depth, adaptive_depth=hp.adaptive_depth,
                               kernel_dim=hp.kernel_dim, hidden=hp.interpolator.hidden_dim, pos_weight=hp.solver.pos_weight,
                               normal_weight=hp.solver.normal_weight, tol=3e-7, net_params=P, timing=tm)
    print('oracle solve:', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in tm.items()}, flush=True)
    out = {'chunk': np.int64(c), 'points': np.int64(xs.shape[0]), 'scale': np.float64(scale), 'shift': sh.astype(np.float32),
           'alpha': fld['alpha'].astype(np.float32), 'iters': np.int64(fld['iters']), 'rel': np.float64(fld['rel']),
           'level_n': np.asarray([L.n for L in fld['hier'].levels], np.int64),
           'level_key_sha256': np.asarray([digest(L.keys.astype(np.int64)) for L in fld['hier'].levels])}
    # field probes: input points of the chunk and the same points moved half a finest voxel along +- their normals
    # (f ~ 0 on the data says little about the scale of f; half a voxel away it is O(gradient x 0.05))
    rs = np.random.RandomState(0)
    pick = np.sort(rs.choice(xs.shape[0], 1500, replace=False))
    off = np.float32(0.5 * hp.voxel_size)
    pxyz = np.concatenate([xs[pick], (xs[pick] + off * nrm[pick]).astype(np.float32), (xs[pick] - off * nrm[pick]).astype(np.float32)]).astype(np.float32)
    f, g = pipeline.evaluate(fld, pxyz, grad=True)
    out['probe_xyz'], out['probe_f'], out['probe_grad'] = pxyz, f.astype(np.float32), g.astype(np.float32)
    out['alpha_absmax'] = np.float64(np.abs(fld['alpha']).max())
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, 'scene_chunk%d_golden.npz' % c)
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f MB) in %.0f s' % (path, os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 27)
