"""ext.sdfgen.sdf_from_points -- ours (csrc/knn.hip: one octree-kNN kernel) against the REFERENCE'S OWN extension (oracle/_ref/nksr_sdfgen.so:
kd-tree build + kNN + estimator, compiled from /root/reference/ext by oracle/build_ref.py) on the same MI355X, same inputs: the one
operation of this project whose reference implementation runs here.  TEST infrastructure (it lives under tests/ because it loads oracle/_ref).
python tests/sdfgen_vs_ref.py [--variants] [--breakdown]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ext  # noqa: E402
from oracle import build_ref  # noqa: E402


def cloud(n, seed=0, noise=0.002):
    rs = np.random.RandomState(seed)
    v = rs.randn(n, 3)
    nrm = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return (nrm * 0.5 + rs.randn(n, 3) * noise).astype(np.float32), nrm


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


VARIANTS = (('octree', {}), ('octree, 2 rings', {'NKSR_SDFGEN_RINGS': '2'}), ('octree, leaf 24', {'NKSR_SDFGEN_LEAF': '24'}),
            ('single grids, x4 per round', {'NKSR_SDFGEN_SEARCH': 'rounds'}))


def main():
    ref = build_ref.load()
    dev = torch.device('cuda:0')
    variants = VARIANTS if '--variants' in sys.argv else VARIANTS[:1]
    print('| reference points | queries | mode | reference ext ms | ' + ' | '.join('nksr_amd ms (%s)' % v[0] for v in variants) +
          ' | speed-up | max abs difference |')
    print('|---|---|---|---|' + '---|' * len(variants) + '---|---|')
    for n_ref, n_q in ((4000, 1000), (20000, 6600), (200000, 200000), (1000000, 1000000)):
        xyz, nrm = cloud(n_ref)
        rs = np.random.RandomState(1)
        q = (xyz[rs.randint(0, n_ref, n_q)] + nrm[rs.randint(0, n_ref, n_q)] * rs.randn(n_q, 1).astype(np.float32) * 0.03).astype(np.float32)
        X, N, Q = [torch.from_numpy(a).to(dev) for a in (xyz, nrm, q)]
        for mode, kw in (('vote k=8', dict(nb_points=8, stdv=0.02)), ('imls k=8', dict(nb_points=8, stdv=0.05, imls=True)),
                         ('vote k=8, adaptive 8', dict(nb_points=8, stdv=3.0, adaptive_knn=8))):
            ours = []
            for _, env in variants:
                os.environ.update(env)
                ours.append(timed(lambda: ext.sdfgen.sdf_from_points(Q, X, N, compute_grad=True, **kw)))
                for key in env:
                    del os.environ[key]
            t_ours, o = ours[0]
            cols = ' | '.join('%.2f' % t for t, _ in ours)
            if ref is not None:
                t_ref, r = timed(lambda: ref.sdf_from_points(Q, X, N, kw['nb_points'], kw['stdv'], True, kw.get('imls', False), kw.get('adaptive_knn', 0)))
                diff = float((o[0] - r[0]).abs().max())
                print('| %d | %d | %s | %.2f | %s | %.1fx | %.1e |' % (n_ref, n_q, mode, t_ref, cols, t_ref / t_ours, diff))
            else:
                print('| %d | %d | %s | (no oracle/_ref) | %s | | |' % (n_ref, n_q, mode, cols))
    if '--breakdown' in sys.argv:
        from nksr_amd._lib import call, ptr, stream
        from nksr_amd.normals import PointGrid, PointPyramid, choose_cell_size
        print()
        print('| reference points = queries | cell size ms | grid ms | octree levels | octree ms | search + estimator kernel ms | whole call ms |')
        print('|---|---|---|---|---|---|---|')
        for n_ref in (4000, 200000, 1000000):
            xyz, nrm = cloud(n_ref)
            rs = np.random.RandomState(1)
            q = (xyz[rs.randint(0, n_ref, n_ref)] + nrm[rs.randint(0, n_ref, n_ref)] * rs.randn(n_ref, 1).astype(np.float32) * 0.03).astype(np.float32)
            X, N, Q = [torch.from_numpy(a).to(dev) for a in (xyz, nrm, q)]
            t_c, cell = timed(lambda: choose_cell_size(X, 8))
            t_g, pg = timed(lambda: PointGrid(X, cell))
            t_p, pyr = timed(lambda: PointPyramid(pg))
            ns = N[pg.perm].contiguous()
            s_, g_, v_ = torch.empty(n_ref, device=dev), torch.empty((n_ref, 3), device=dev), torch.empty(n_ref, dtype=torch.int32, device=dev)
            t_k, _ = timed(lambda: call('nksr_sdf_from_points_pyramid', pyr.struct, ptr(ns), None, ptr(Q), n_ref, 8, 4, 0.02, 0, ptr(s_), ptr(g_),
                                        ptr(v_), stream()))
            t_all, _ = timed(lambda: ext.sdfgen.sdf_from_points(Q, X, N, 8, 0.02, True))
            print('| %d | %.2f | %.2f | %d | %.2f | %.2f | %.2f |' % (n_ref, t_c, t_g, pyr.levels, t_p, t_k, t_all))


if __name__ == '__main__':
    main()
