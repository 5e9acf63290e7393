"""Edge cases of the hot path on the GPU: empty / tiny / ragged / degenerate / out-of-range inputs,
hash collisions, error behaviour (RuntimeError, never abort -- SURVEY.md section 5)."""
import numpy as np
import pytest
import torch

from conftest import make_cloud

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def test_input_validation_raises_runtime_error():
    import nksr
    rec = nksr.Reconstructor(DEV)
    xyz, nrm = make_cloud('sphere', 500, 0.0, 0)
    x, n = torch.from_numpy(xyz).to(DEV), torch.from_numpy(nrm).to(DEV)
    with pytest.raises(RuntimeError):
        rec.reconstruct(x.double(), n)                         # non-float32 input must error
    with pytest.raises(RuntimeError):
        rec.reconstruct(x[:, :2].contiguous(), n)              # wrong shape
    with pytest.raises(RuntimeError):
        rec.reconstruct(x)                                     # no orientation information at all
    with pytest.raises(RuntimeError):
        rec.reconstruct(x[:5], n[:5], detail_level=None)       # fewer than 8 points
    bad = x.clone()
    bad[3, 1] = float('nan')
    with pytest.raises(RuntimeError):
        rec.reconstruct(bad, n, detail_level=None)
    with pytest.raises(RuntimeError):
        rec.reconstruct(x * 1e7, n, detail_level=None)         # beyond the 2^20-voxel key range
    with pytest.raises(RuntimeError):
        rec.reconstruct(x, n, chunk_size=1.0, voxel_size=0.05)  # NKSR-USAGE.md:137
    # the session is still usable afterwards
    f = rec.reconstruct(x, n, voxel_size=0.08)
    assert f.extract_dual_mesh().f.shape[0] > 0


def test_duplicates_boundaries_and_negative_coordinates():
    """Many coincident points, points exactly on voxel faces / corners, negative coordinates:
    integer decisions still agree with the oracle bit for bit."""
    import nksr_amd
    from oracle import hierarchy
    rs = np.random.RandomState(0)
    lattice = (rs.randint(-40, 40, size=(3000, 3)).astype(np.float32)) * np.float32(0.05)   # on faces / corners
    dup = np.repeat(lattice[:50], 20, 0)
    xyz = np.concatenate([lattice, dup, -np.abs(rs.randn(500, 3).astype(np.float32))])
    oh = hierarchy.Hierarchy(0.1, 4).build_point_neighborhood(xyz)
    oe = hierarchy.Hierarchy(0.1, 4).build_point_splatting(xyz)
    x = torch.from_numpy(xyz).to(DEV)
    gh = nksr_amd.SparseFeatureHierarchy(0.1, 4, DEV).build_point_neighborhood(x)
    ge = nksr_amd.SparseFeatureHierarchy(0.1, 4, DEV).build_point_splatting(x)
    for d in range(4):
        assert np.array_equal(gh.level(d).keys.cpu().numpy(), oh.levels[d].keys)
        assert np.array_equal(ge.level(d).keys.cpu().numpy(), oe.levels[d].keys)
        assert np.array_equal(gh.level(d).nbr.cpu().numpy(), oh.levels[d].nbr)


def test_hash_table_under_collisions_and_misses():
    from nksr_amd import ops
    # keys that agree in their low bits (stress linear probing) + a dense run
    base = torch.arange(0, 20000, dtype=torch.int64, device=DEV)
    keys = torch.cat([base * (1 << 20), base + 7]).unique()
    keys = ops.sort_keys(keys.contiguous())
    h = ops.HashTable(keys)
    got = h.query(keys)
    assert torch.equal(got.long(), torch.arange(keys.numel(), device=DEV))
    miss = h.query((keys + (1 << 40)).contiguous())
    assert bool((miss == -1).all())
    assert h.query(torch.empty(0, dtype=torch.int64, device=DEV)).numel() == 0


def test_tiny_clouds_and_single_voxel_systems():
    import nksr
    rec = nksr.Reconstructor(DEV)
    # 8 points in one voxel: smallest legal input; must solve and evaluate without NaN
    p = (torch.rand(8, 3, device=DEV) * 0.05)
    n = torch.nn.functional.normalize(torch.randn(8, 3, device=DEV), dim=1)
    f = rec.reconstruct(p, n, detail_level=None)
    v = f.evaluate_f(p, grad=True)
    assert torch.isfinite(v.value).all() and torch.isfinite(v.gradient).all()
    assert f.solve_info['rel_residual'] <= 1e-5
    m = f.extract_dual_mesh(mise_iter=2)
    assert m.v.shape[1] == 3 and m.f.dtype == torch.int64
    # evaluation far away from everything: exactly zero, no fault
    far = torch.full((4, 3), 500.0, device=DEV)
    assert float(f.evaluate_f(far).value.abs().max()) == 0.0
    assert f.evaluate_f(torch.zeros((0, 3), device=DEV)).value.numel() == 0


def test_compaction_and_scan_primitives():
    from nksr_amd import ops
    for n in (0, 1, 255, 256, 257, 100003):
        fl = (torch.rand(n, device=DEV) < 0.37).to(torch.int32)
        sel = ops.compact(fl)
        assert torch.equal(sel.long(), torch.nonzero(fl).reshape(-1))
        ex = ops.exclusive_sum_i32(fl)
        ref = torch.cumsum(fl, 0) - fl
        assert torch.equal(ex, ref.to(torch.int32))
    k = torch.randint(0, 1 << 50, (200001,), dtype=torch.int64, device=DEV)
    assert torch.equal(ops.sort_unique(k), torch.unique(k))


def test_grid_upsample_and_max_points_batches():
    import nksr
    rec = nksr.Reconstructor(DEV)
    xyz, nrm = make_cloud('sphere', 3000, 0.002, 0)
    f = rec.reconstruct(torch.from_numpy(xyz).to(DEV), torch.from_numpy(nrm).to(DEV), voxel_size=0.06)
    a = f.extract_dual_mesh(grid_upsample=2)
    b = f.extract_dual_mesh(mise_iter=1)
    # upsampling the base grid by 2 visits a superset of the MISE(1) cells: same lattice, >= triangles
    assert a.f.shape[0] >= b.f.shape[0] > 0
    c = f.extract_dual_mesh(mise_iter=1, max_points=1000)      # tiny evaluation batches
    assert torch.equal(b.f, c.f) and torch.equal(b.v, c.v)
    r = np.linalg.norm(a.v.cpu().numpy(), axis=1)
    assert abs(np.median(r) - 0.45) < 0.01


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's keys + roofline + cpu_baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
                          '--points', '60000', '--cpu-sample', '1500', '--scene-points', '640000', '--cpu-cores', '2', '--cpu-repeats', '1'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 1 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert 'workload' in d['config'] and d['value'] > 0
    # configs[4] is the top-level workload at every N (the curve's N = 1 point), the same scene = strong scaling
    assert 'configs[4]' in d['config']['workload'] and d['scaling'] == 'strong' and d['config']['chunks'] == 64 and d['config']['tree_depth'] == 5
    assert d['dist']['world_size'] == 1 and d['dist']['rccl_ranks_seen'] == 1 and 'device_allocs_per_step' in d['allocator']
    assert set(d['stages_s_per_step']) >= {'t_network', 't_assemble', 't_pcg', 't_mesh'}
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    for k in ('traffic', 'traffic_source', 'bytes_per_launch', 'avg_launch_us', 'launches_timed', 'kernel'):
        assert k in r, k
    assert (r['traffic'] is None) == (r['traffic_source'] is None) and r['launches_timed'] > 0
    # roofline.traffic is collected in the run itself when rocprofv3 is on the box (two counter passes over one extra step)
    import shutil
    if shutil.which('rocprofv3'):
        # (a profiler that cannot collect on this box must not fail the measurement: the line then says why and falls back)
        assert (r['traffic_source'] and r['traffic_source'].startswith('live') and r['traffic'] > 0.5 * r['bytes_per_launch']) or r.get('traffic_live_note'), r
    # a roofline fraction is a fraction: algorithmic minimum <= what the layout moves <= what the HBM could stream
    assert d['config']['fused_mode'] is True and 'k_fz_cells' in r['kernel'] and 0 < r['frac'] < r['frac_physical'] <= 1.0
    c2 = d['cloud_1m']                 # configs[2]: one field, the operator roofline + the assembled solve's CSR SpMV roofline
    assert 'configs[2]' in c2['config']['workload'] and c2['value'] > 0
    rc = c2['roofline']
    # algorithmic minimum of the matrix-free operator: 4 B per stored entry of G and Q (counted on the device) + 4 B per row and
    # level (row -> cell) + 116 B per unknown (stencil, x, y); SURVEY.md section 8d's formula (16 B per stored entry) beside it
    se, slots, M = c2['config']['stored_entries_G_Q'], c2['config']['kernel_row_slots'], c2['config']['unknowns_M']
    assert 0 < se <= slots and abs(rc['bytes_per_launch'] - (4.0 * se + 4.0 * slots / 27 + 116.0 * M + 4)) < 1.0
    assert abs(rc['survey_formula_bytes_per_launch'] - (16.0 * se + 12.0 * M + 4)) < 1.0
    assert d['spmv_csr_roofline']['kernel'].startswith('k_spmv') and 0 < d['spmv_csr_roofline']['frac'] <= 1.0
    o = c2['other_solve_mode']          # the assembled CSR solve on the same workload, with the CSR SpMV roofline
    assert o['fused_mode'] is False and o['value'] > 0 and 'k_spmv' in o['roofline']['kernel'] and o['nnz_A'] > 0
    assert 'achieved_physical' in o['roofline'] and o['roofline']['frac_physical'] <= o['roofline']['frac']
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 2 and c['value'] > 0 and 'recons_waymo_cpu.py' in c['sample']
    assert c['gpu_same_input']['value'] > 0 and c['gpu_same_input']['ms'] > 0 and c['workload_crop']['value'] > 0
    assert d['small_inputs']['configs1_shapenet_3k']['ms'] > 0


def test_reconstruct_is_bitwise_deterministic():
    """No float atomics anywhere on the path (integer atomics only, fixed reduction orders): two runs on the
    same input give bit-identical coefficients, matrix and mesh."""
    import nksr_amd
    from nksr_amd import utils
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(120000, seed=3, extent=(12.0, 12.0, 6.0), n_objects=4)
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev)
    for fused in (False, True):         # assembled CSR solve / matrix-free solve
        runs = []
        for _ in range(2):
            fld = rec.reconstruct(xyz, nrm, detail_level=1.0, fused_mode=fused)
            mesh = fld.extract_dual_mesh(mise_iter=1)
            mat = (fld.matrix[1].clone(), fld.matrix[2].clone()) if not fused else (fld.diag.clone(), fld.diag.clone())
            runs.append((fld.alpha.clone(), mat[0], mat[1], fld.rhs.clone(), mesh.v.clone(), mesh.f.clone(), fld.solve_info['iters']))
        a, b = runs
        assert a[6] == b[6]
        for u, v in zip(a[:6], b[:6]):
            assert torch.equal(u, v)


@pytest.mark.parametrize('drop', [0.0, 0.3])
def test_neighbour_table_from_the_parent_level_equals_the_hashed_one_also_when_parents_are_missing(drop):
    """csrc/hierarchy.hip: a level's 27-neighbour table is derived from the next-coarser table and its voxels' children
    (k_build_nbr_parent); a key list whose voxels lack parents (drop > 0: a foreign list, no hierarchy builder makes one)
    takes the hash path on the device (k_build_nbr_orphans). Either way the table is the one 27 hash probes per voxel give."""
    from nksr_amd import ops
    from nksr_amd._lib import call, ptr, stream
    from nksr_amd.svh import SparseGrid
    g = torch.Generator(device='cpu').manual_seed(11)
    blob = (torch.randn(60000, 3, generator=g) * 9).round().to(torch.int32)
    ijk = torch.cat([blob, torch.tensor([[-40, 3, 7], [41, 41, -41]], dtype=torch.int32)]).to(DEV).contiguous()
    keys = torch.empty(ijk.shape[0], dtype=torch.int64, device=DEV)
    call('nksr_encode_keys', ptr(ijk), ijk.shape[0], 0, ptr(keys), stream())
    keys = ops.sort_unique(keys)
    pk = ops.sort_unique((keys >> 3).contiguous())
    if drop:
        keep = torch.rand(pk.numel(), generator=g) >= drop
        pk = pk[keep.to(DEV)].contiguous()
    coarse = SparseGrid(pk, 1, 0.1)
    want = SparseGrid(keys, 0, 0.1).nbr                       # 27 hash probes per voxel
    got = SparseGrid(keys, 0, 0.1, coarse=coarse).nbr
    assert want.shape[0] > 20000 and int((want >= 0).sum()) > 3 * want.shape[0]
    assert torch.equal(got, want)
