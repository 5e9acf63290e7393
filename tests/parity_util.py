"""Helpers of the GPU parity tests: unconditional topology comparison and measured-error reporting.

Topology bar (BASELINE.json north_star: "index-exact for voxel/triangle topology"):
the HIP mesh and the oracle mesh must hold the SAME triangles -- compared through the canonical identity
of their vertices, (lattice key of the lower end point of the crossed lattice edge, axis), so vertex
numbering does not matter -- EXCEPT inside cells where fp32 noise can legitimately flip a sign decision.
That set is computed on the ORACLE side only:
  * level m of the MISE refinement: a lattice vertex is *ambiguous* when |f| < eps, a cell is ambiguous when one
    of its 8 corners is; ambiguity spreads to the 26 neighbouring cells (the hanging-vertex rule couples a cell's
    refined face values to whether its neighbour was refined) and to every descendant at finer levels;
  * eps = 10 x the measured max |f_hip - f_oracle| over the oracle's lattice vertices (itself asserted <= the
    1e-4 contract of SURVEY.md section 8c), so the test calibrates itself and cannot hide a regression.
Every differing triangle must lie in a tainted cell; the tainted fraction is bounded and printed.
"""
import numpy as np

from oracle import meshing as omesh


def report(name, **kv):
    """One line per test with the measured errors (pytest -s / captured output shows it on failure, and
    tests/ collects them in gpurun_out/parity_report.txt when NKSR_PARITY_REPORT is set)."""
    import os
    line = '[parity] %s: ' % name + ' '.join('%s=%s' % (k, ('%.3e' % v) if isinstance(v, float) else v) for k, v in kv.items())
    print(line)
    path = os.environ.get('NKSR_PARITY_REPORT')
    if path:
        with open(path, 'a') as f:
            f.write(line + '\n')


def check(name, measured, bound):
    """Tolerance check that always REPORTS the measured value next to its bound.  NKSR_PARITY_CALIBRATE=1 turns the
    assertion off (used once per round on the GPU box to read all measured errors in one pass)."""
    import os
    measured, bound = float(measured), float(bound)
    report(name, measured=measured, bound=bound, ok=bool(measured <= bound))
    if not os.environ.get('NKSR_PARITY_CALIBRATE'):
        assert measured <= bound, '%s: measured %.3e > bound %.3e' % (name, measured, bound)


def check_alpha(name, alpha_hip, ofl, tol):
    """Converged coefficient vectors.  The multi-level basis is redundant (a coarse B-spline is a combination of fine
    ones wherever the fine level exists), so A has near-null directions held only by reg*I: two solves that both reach
    a relative RESIDUAL of ``tol`` agree in alpha only to cond(A) x tol, while the field they define agrees to ~tol.
    What is checked, therefore: (1) the HIP alpha solves the ORACLE's system to 10 x tol (fp64 residual with the oracle's
    own matrix and right-hand side) -- the conditioning-free statement of "same solution"; (2) alpha itself within
    ALPHA_TOL = 1e-4 of the oracle's, the contract of SURVEY.md section 8c (round 2 needed 2e-3: 4.2e-4 measured; since the
    fp32 rounding fixes of round 3 the worst case over all cases is 4.2e-5).  The same 1e-4 holds after a FIXED number of
    iterations (tests/test_gpu_parity.py::test_pcg_matches_oracle_and_scipy) and on the field values (every test here)."""
    a = np.asarray(alpha_hip, np.float64)
    A, b = ofl['A'].astype(np.float64), ofl['b'].astype(np.float64)
    check(name + ':residual_in_oracle_system', np.linalg.norm(b - A @ a) / np.linalg.norm(b), 10.0 * tol)
    check(name + ':alpha_rel', np.abs(a - ofl['alpha']).max() / np.abs(ofl['alpha']).max(), ALPHA_TOL)


ALPHA_TOL = 1e-4      # SURVEY.md section 8c / BASELINE.md section 2.1 (measured on MI355X: <= 4.2e-5 over all cases, profiles/r03_parity_report.txt)


def _rows_view(a):
    a = np.ascontiguousarray(a)
    return a.view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).ravel()


def canonical_triangles(faces, vkey, axis):
    """[T, 3] vertex indices -> [T, 6] canonical ids (3 lattice keys, 3 axes), corner order preserved.
    (Lattice keys are 63-bit Morton codes: key and axis cannot share one int64.)"""
    f = faces.astype(np.int64)
    return np.concatenate([vkey.astype(np.int64)[f], axis.astype(np.int64)[f]], 1)


def triangle_cells(tri_ids):
    """Lattice cell(s) of triangles given by canonical vertex ids.  An edge (lower vertex g, axis a) lies in the
    cells c with c[a] == g[a] and c[b] in {g[b] - 1, g[b]} for b != a; the three edges of a marching-cubes
    triangle pin the cell, unless all three lie in one lattice face (then both cells sharing it qualify).
    Returns (lo [T,3], hi [T,3]) inclusive candidate ranges (lo == hi when unique)."""
    vk, ax = tri_ids[:, :3], tri_ids[:, 3:]
    g = omesh.lattice_decode(vk.reshape(-1)).astype(np.int64).reshape(vk.shape + (3,))     # [T,3 corners,3 xyz]
    lo = np.full((tri_ids.shape[0], 3), -(1 << 40), np.int64)
    hi = np.full((tri_ids.shape[0], 3), (1 << 40), np.int64)
    for k in range(3):
        for b in range(3):
            on_axis = ax[:, k] == b
            l = np.where(on_axis, g[:, k, b], g[:, k, b] - 1)
            h = g[:, k, b]
            lo[:, b] = np.maximum(lo[:, b], l)
            hi[:, b] = np.minimum(hi[:, b], h)
    return lo, hi


def ref_from_info(ov, of, info):
    """Reference mesh record from a live oracle run (oracle.meshing.extract(..., info=))."""
    levels = []
    for L in info['levels']:
        cmin = np.abs(L['f'])[L['cidx']].min(1) if len(L['cidx']) else np.zeros(0, np.float32)
        levels.append((L['cells'].astype(np.int64), cmin, len(L['cells'])))
    return {'v': ov, 'f': of, 'vert_vkey': info['vert_vkey'], 'vert_axis': info['vert_axis'], 'vert_df': info['vert_df'],
            'h': float(info['h']), 'levels': levels,
            'probes': [(L['pos'], L['f_raw']) for L in info['levels']]}


def ref_from_golden(g, prefix='mesh_', w0=0.1):
    """Same record from a committed fixture (oracle/make_golden_chunked.py: only the near-threshold cells and a
    sample of the lattice vertices are stored)."""
    nl = int(g[prefix + 'nlevels'])
    levels = [(g[prefix + 'near_cells_%d' % m].astype(np.int64), g[prefix + 'near_minabs_%d' % m], int(g[prefix + 'ncells_%d' % m]))
              for m in range(nl)]
    probes = []
    for m in range(nl):
        gk = omesh.lattice_decode(g[prefix + 'probe_vk_%d' % m])
        probes.append((omesh.lattice_positions(gk, float(g[prefix + 'lat_h_%d' % m]), 0.5 * w0), g[prefix + 'probe_f_%d' % m]))
    return {'v': g[prefix + 'v'], 'f': g[prefix + 'f'], 'vert_vkey': g[prefix + 'vert_vkey'], 'vert_axis': g[prefix + 'vert_axis'],
            'vert_df': g[prefix + 'vert_df'], 'h': float(g[prefix + 'h']), 'levels': levels, 'probes': probes}


def tainted_cells(levels, eps):
    """Per MISE level: sorted lattice keys of the tainted cells at that level (see module docstring).
    ``levels``: [(cell coords [k,3], min |f| over the cell's corners [k], number of cells)]."""
    out = []
    carried = None              # tainted cells of the previous level, as coords
    for cells, cmin, _ in levels:
        seeds = cells[cmin < eps]
        if len(seeds):
            off = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], np.int64)
            seeds = (seeds[:, None, :] + off[None]).reshape(-1, 3)
        if carried is not None and len(carried):
            ch = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.int64)
            kids = (carried[:, None, :] * 2 + ch[None]).reshape(-1, 3)
            seeds = np.concatenate([seeds, kids]) if len(seeds) else kids
        keys = np.unique(omesh.lattice_key(seeds)) if len(seeds) else np.zeros(0, np.int64)
        out.append(keys)
        carried = omesh.lattice_decode(keys).astype(np.int64) if len(keys) else None
    return out


def compare_meshes(name, gv, gf, gvkey, gaxis, ref, delta_f, w0=0.1, max_tainted_frac=0.06, eps=None, max_over_plain_frac=1e-3):
    """Unconditional topology + vertex-position comparison (positions in model units) of a HIP mesh against a
    reference record (ref_from_info / ref_from_golden).  ``delta_f``: measured max |f_hip - f_oracle| at the
    oracle's lattice vertices.  Returns a stats dict."""
    eps = 10.0 * float(delta_f) if eps is None else float(eps)
    ov, of = ref['v'], ref['f']
    taint = tainted_cells(ref['levels'], eps)[-1]
    n_final = ref['levels'][-1][2]
    n_taint = len(taint)            # counts dilated (possibly non-existing) cells too: an over-estimate
    tg = canonical_triangles(gf, gvkey, gaxis)
    to = canonical_triangles(of, ref['vert_vkey'], ref['vert_axis'])
    vg, vo = _rows_view(tg), _rows_view(to)
    only_g = ~np.isin(vg, vo)
    only_o = ~np.isin(vo, vg)
    bad = 0
    for tri in (tg[only_g], to[only_o]):
        if len(tri):
            lo, hi = triangle_cells(tri)
            ok = np.isin(omesh.lattice_key(lo), taint) | np.isin(omesh.lattice_key(hi), taint)
            bad += int((~ok).sum())
    # vertex positions of the common vertices
    idg = _rows_view(np.stack([gvkey.astype(np.int64), gaxis.astype(np.int64)], 1))
    ido = _rows_view(np.stack([ref['vert_vkey'].astype(np.int64), ref['vert_axis'].astype(np.int64)], 1))
    common, ig, io = np.intersect1d(idg, ido, return_indices=True)
    dv = np.abs(gv[ig].astype(np.float64) - ov[io].astype(np.float64)).max(1) if len(common) else np.zeros(0)
    h = ref['h']
    bound = np.maximum(1e-4 * w0, 2.0 * h * float(delta_f) / np.maximum(ref['vert_df'][io].astype(np.float64), 1e-30))
    n_over_plain = int((dv > 1e-4 * w0).sum())
    stats = {'T_hip': len(tg), 'T_oracle': len(to), 'only_hip': int(only_g.sum()), 'only_oracle': int(only_o.sum()),
             'outside_tainted': bad, 'tainted_cells': n_taint, 'final_cells': n_final, 'eps': eps,
             'common_vertices': len(common), 'max_dv_voxel': float(dv.max() / w0) if len(dv) else 0.0,
             'vertices_over_1e-4_voxel': n_over_plain}
    report(name, **stats)
    import os
    if os.environ.get('NKSR_PARITY_CALIBRATE'):
        stats['exact'] = not only_g.any() and not only_o.any()
        return stats
    assert bad == 0, '%s: %d differing triangles lie outside the near-threshold cells' % (name, bad)
    assert n_taint <= max(64, max_tainted_frac * n_final), '%s: tainted set too large (%d of %d cells)' % (name, n_taint, n_final)
    exact = not only_g.any() and not only_o.any()
    if exact:   # identical triangle sets: the ORDER must be identical too (cell-major, table order) => index-exact faces
        assert np.array_equal(tg, to), '%s: same triangles in a different order' % name
        assert np.array_equal(np.asarray(gf, np.int64), np.asarray(of, np.int64)), '%s: face indices differ' % name
    # vertices beyond the PLAIN contract (1e-4 voxel) -- allowed only through the widened bound above (an edge
    # whose end values nearly coincide amplifies the field's own 1e-6 difference): tracked, and capped at one vertex in a thousand
    # (round 4 measured 14 / 42 282 on street8, 27 / 65 386 on terrain5, <= 3 elsewhere)
    if max_over_plain_frac is not None:
        assert n_over_plain <= max(2, max_over_plain_frac * len(common)), '%s: %d of %d vertices beyond 1e-4 voxel' % (name, n_over_plain, len(common))
    if not (dv <= bound).all():
        w = int(np.argmax(dv / bound))
        raise AssertionError('%s: %d vertices beyond their bound; worst: off by %.3e voxel, bound %.3e voxel, |f0 - f1| = %.3e, at %s (oracle %s)' % (
            name, int((dv > bound).sum()), dv[w] / w0, bound[w] / w0, float(ref['vert_df'][io][w]), gv[ig][w], ov[io][w]))
    stats['exact'] = exact
    return stats


def lattice_delta(field_eval, ref):
    """max |f_hip - f_oracle| over the oracle's lattice vertices of every MISE level (raw evaluations, before the
    hanging-vertex rule) and the largest |f| there.  ``field_eval(pos [n,3] f32 model units) -> f [n] np``."""
    d, fmax = 0.0, 0.0
    for pos, f_raw in ref['probes']:
        if len(pos) == 0:
            continue
        fg = field_eval(np.ascontiguousarray(pos, np.float32))
        d = max(d, float(np.abs(fg - f_raw).max()))
        fmax = max(fmax, float(np.abs(f_raw).max()))
    return d, fmax


def mesh_arrays(mesh, scale=1.0):
    """(v in model units, f, edge_vkey, edge_axis) numpy arrays of a nksr_amd MeshingResult."""
    return (mesh.v.cpu().numpy() * np.float32(scale), mesh.f.cpu().numpy(), mesh.edge_vkey.cpu().numpy(),
            mesh.edge_axis.cpu().numpy().astype(np.int64))


def assert_closed(faces, name='mesh'):
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), 1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all(), '%s is not closed' % name
    return len(cnt)


def mesh_parity(name, fld, ofl, mise_iter, scale=1.0, grid_upsample=1, w0=0.1):
    """HIP field ``fld`` (KernelField, global scale ``scale``) against the live oracle field ``ofl`` (oracle.pipeline dict):
    lattice field values within 1e-4 of max|f|, then the unconditional mesh comparison."""
    import torch
    from oracle import pipeline
    info = {}
    ov, of = pipeline.extract_dual_mesh(ofl, mise_iter=mise_iter, grid_upsample=grid_upsample, info=info)
    ref = ref_from_info(ov, of, info)
    dev = fld.device
    ev = lambda p: fld._evaluate_f_model(torch.from_numpy(p).to(dev), False).value.cpu().numpy()
    delta, fmax = lattice_delta(ev, ref)
    check(name + ':lattice_f_rel', delta / max(fmax, 1e-30), 1e-4)
    mesh = fld.extract_dual_mesh(mise_iter=mise_iter, grid_upsample=grid_upsample)
    st = compare_meshes(name, *mesh_arrays(mesh, scale), ref, delta_f=delta, w0=w0)
    return st, mesh, (ov, of)


def window_mesh_check(name, fld, scale, mise_iter, centre_ijk, radius, w0=0.1, min_triangles=200):
    """Bench-scale meshes against the oracle where the oracle can go: a WINDOW of the finest level.

    The voxels of ``fld.svh.level(0)`` inside the box centre +- radius become an oracle level; oracle.meshing.extract runs on it
    with the HIP field's own lattice values (``_evaluate_f_model`` at the oracle's lattice positions -- the same positions, the same
    kernel: the comparison isolates the integer / topology work and the vertex interpolation at the size bench.py times) and the
    HIP mask.  Base cells at least two cells inside the box see exactly the neighbourhood they have in the full mesh (MISE's
    hanging-vertex rule looks one cell around); every oracle triangle of those cells must be in the HIP mesh, in the same order,
    every HIP triangle whose cell lies three cells inside must be in the oracle's, and common vertices agree to 1e-4 voxel (the
    plain SURVEY.md section 8d bound -- no widening: both sides interpolate the same values)."""
    import torch
    from oracle import hierarchy as ohier
    dev = fld.svh.level(0).ijk.device
    g0 = fld.svh.level(0)
    ijk = g0.ijk.cpu().numpy().astype(np.int64)
    keys = g0.keys.cpu().numpy()
    c = np.asarray(centre_ijk, np.int64)
    lo, hi = c - radius, c + radius
    inw = ((ijk >= lo[None]) & (ijk <= hi[None])).all(1)
    lvl = ohier.Level(keys[inw], 0, w0)
    lvl.build_nbr()

    def ev(p):
        t = torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(dev)
        return fld._evaluate_f_model(t, False, max_points=1 << 22).value.cpu().numpy()

    def mk(p):
        m = fld.mask_vertices(torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(dev))
        return np.ones(len(p), bool) if m is None else m.cpu().numpy()
    info = {}
    ov, of = omesh.extract(w0, lvl, ev, mise_iter=mise_iter, grid_upsample=1, mask_fn=mk, info=info)
    to = canonical_triangles(of, info['vert_vkey'], info['vert_axis'])
    base_o = info['tri_cell'].astype(np.int64) >> mise_iter
    in2 = ((base_o >= (lo + 2)[None]) & (base_o <= (hi - 3)[None])).all(1)          # (a base cell spans ijk .. ijk + 1)
    # the HIP mesh, cut down on the device to the triangles around the window before anything moves to the host
    mesh = fld.extract_dual_mesh(mise_iter=mise_iter)
    vm = mesh.v * float(scale)
    blo = torch.tensor(((lo - 1) * w0).tolist(), dtype=torch.float32, device=vm.device)
    bhi = torch.tensor(((hi + 2) * w0).tolist(), dtype=torch.float32, device=vm.device)
    vin = ((vm >= blo[None]) & (vm <= bhi[None])).all(1)
    tin = torch.nonzero(vin[mesh.f].all(1)).reshape(-1)
    gf = mesh.f[tin].cpu().numpy()
    uv, inv = np.unique(gf.reshape(-1), return_inverse=True)
    gf = inv.reshape(-1, 3)
    sel = torch.from_numpy(uv).to(vm.device)
    gv, gk, ga = vm[sel].cpu().numpy(), mesh.edge_vkey[sel].cpu().numpy(), mesh.edge_axis[sel].cpu().numpy().astype(np.int64)
    tg = canonical_triangles(gf, gk, ga)
    vg, vo = _rows_view(tg), _rows_view(to)
    pos_in_hip = {k.tobytes(): i for i, k in enumerate(vg)}
    where = np.asarray([pos_in_hip.get(k.tobytes(), -1) for k in vo[in2]], np.int64)
    missing = int((where < 0).sum())
    ordered = bool((np.diff(where[where >= 0]) > 0).all())
    clo, chi = triangle_cells(tg)
    in3 = (((clo >> mise_iter) >= (lo + 3)[None]) & ((chi >> mise_iter) <= (hi - 4)[None])).all(1)
    extra = int((~np.isin(vg[in3], vo)).sum())
    idg = _rows_view(np.stack([gk.astype(np.int64), ga], 1))
    ido = _rows_view(np.stack([info['vert_vkey'].astype(np.int64), info['vert_axis'].astype(np.int64)], 1))
    inner_v = np.zeros(len(ido), bool)                       # vertices of the compared triangles only: at the window's rim the oracle's
    inner_v[np.asarray(of)[in2].reshape(-1)] = True          # hanging-vertex rule sees cells the full mesh has and the window lacks
    _, ig, io = np.intersect1d(idg, ido[inner_v], return_indices=True)
    io = np.nonzero(inner_v)[0][io]
    dv = float(np.abs(gv[ig].astype(np.float64) - ov[io].astype(np.float64)).max() / w0) if len(ig) else 0.0
    report(name, window_voxels=int(inw.sum()), oracle_triangles=len(to), compared=int(in2.sum()), hip_inner=int(in3.sum()), missing=missing,
           extra=extra, ordered=ordered, common_vertices=len(ig), max_dv_voxel=dv)
    assert int(in2.sum()) >= min_triangles, '%s: window holds only %d triangles' % (name, int(in2.sum()))
    assert missing == 0 and extra == 0, '%s: %d oracle triangles missing from the HIP mesh, %d HIP triangles unknown to the oracle' % (name, missing, extra)
    assert ordered, '%s: same triangles in a different order' % name
    check(name + ':vertex_dv_voxel', dv, 1e-4)
    return {'compared': int(in2.sum()), 'max_dv_voxel': dv}
