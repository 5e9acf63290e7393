"""CPU-side contract of bench.py (the driver parses its one JSON line): record keys, the roofline arithmetic, the
chunk -> rank map of the strong-scaling scene and the sharded tile loader's bookkeeping."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_record_arithmetic_and_keys():
    b = _bench()
    # 220 launches of 0.58 ms each moving 3.01 GB (algorithmic) / 2.50 GB (physical)
    r = b.roofline_record(220 * 0.58, 220, 220 * 3.01e9, 220 * 2.50e9, 220 * 3.01e9)
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'achieved_physical', 'frac_physical',
              'bytes_per_launch', 'physical_bytes_per_launch', 'avg_launch_us', 'launches_timed', 'kernel'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['achieved'] - 3.01e9 / 0.58e-3 / 1e9) < 1e-6 * r['achieved']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['frac_physical'] < r['frac'] <= 1.0
    assert abs(r['avg_launch_us'] - 580.0) < 1e-6
    # traffic is only attached when the committed PMC pass was taken on the same matrix, and then names its (static) source
    assert (r['traffic'] is None) == (r['traffic_source'] is None)
    # the matrix-free record: achieved = the operator's ALGORITHMIC MINIMUM / time (a fraction of the peak, <= 1 by construction:
    # it is less than what the layout moves); the SURVEY formula's figure travels beside it, labelled, and may exceed 1
    f = b.roofline_record(160 * 0.56, 160, 160 * 1.60e9, 160 * 2.58e9, 160 * 5.65e9, fused=True)
    assert set(r) | {'survey_formula_bytes_per_launch', 'survey_formula_frac'} == set(f) and 'k_fz_sweep' in f['kernel']
    assert abs(f['achieved'] - 1.60e9 / 0.56e-3 / 1e9) < 1e-6 * f['achieved'] and abs(f['achieved_physical'] - 2.58e9 / 0.56e-3 / 1e9) < 1e-6 * f['achieved']
    assert 0 < f['frac'] < f['frac_physical'] <= 1.0 < f['survey_formula_frac']
    z = b.roofline_record(0.0, 0, 0.0, 0.0, 0.0)
    assert z['achieved'] == 0.0 and z['traffic'] is None


def test_strong_scaling_partition_is_compact_and_balanced():
    from nksr_amd import dist as D
    for world in (1, 2, 4, 8):
        owner = D.partition_chunks(64, world, [156250] * 64, grid=(8, 8, 1))
        assert sorted(set(owner)) == list(range(world))
        for r in range(world):
            mine = [c for c in range(64) if owner[c] == r]
            assert len(mine) == 64 // world
            cx, cy = np.array([c // 8 for c in mine]), np.array([c % 8 for c in mine])
            # a Morton-contiguous piece of 64 / world tiles is an axis-aligned block
            assert (cx.max() - cx.min() + 1) * (cy.max() - cy.min() + 1) == len(mine)
    # weights shift the cuts but keep every piece contiguous along the curve
    w = [1000] * 32 + [10] * 32
    o = D.partition_chunks(64, 4, w, grid=(8, 8, 1))
    load = [sum(w[c] for c in range(64) if o[c] == r) for r in range(4)]
    assert max(load) <= 1.3 * sum(w) / 4


def test_terrain_tiles_are_deterministic_and_exactly_sized():
    from nksr_amd import utils
    a, na = utils.terrain_tile((2, 5), 20000, 125.0, seed=0)
    b, nb = utils.terrain_tile((2, 5), 20000, 125.0, seed=0)
    assert a.shape == (20000, 3) and np.array_equal(a, b) and np.array_equal(na, nb)
    assert a[:, 0].min() >= 250.0 and a[:, 0].max() <= 375.0 and a[:, 1].min() >= 625.0 and a[:, 1].max() <= 750.0
    np.testing.assert_allclose(np.linalg.norm(na, axis=1), 1.0, atol=1e-5)
    c, _ = utils.terrain_tile((2, 6), 20000, 125.0, seed=0)
    assert not np.array_equal(a, c)


def test_committed_pmc_records_feed_the_traffic_field():
    """``roofline.traffic`` is taken from the committed rocprofv3 --pmc passes (static: the bench does not run counters) and only when
    the pass was taken on the same system: the committed bench line names the files it used, they exist, and the loaders find them."""
    import json
    b = _bench()
    line = json.loads(open(os.path.join(ROOT, 'profiles', 'r03_bench.json')).read().strip().splitlines()[-1])
    recs = [('headline', line['roofline'], True), ('csr', line['spmv_csr_roofline'], False), ('scene', line['scale_scene']['roofline'], True)]
    for name, r, fused in recs:
        assert 0.0 < r['frac'] <= 1.0 and r['frac'] <= r['frac_physical'] * 1.3, name
        assert r['traffic'] is not None and r['traffic_source'].startswith('static: profiles/'), name
        src = os.path.join(ROOT, r['traffic_source'][len('static: '):])
        assert os.path.exists(src), src
        got, f = (b.load_traffic_fused(r['physical_bytes_per_launch']) if fused else b.load_traffic(r['bytes_per_launch']))
        assert f is not None and abs(got - r['traffic']) <= 5e-3 * r['traffic'], name      # (the PMC file may be a later pass of the same command)
        # the counters agree with the byte model the physical fraction is priced on: no hidden re-reads
        model = r['physical_bytes_per_launch']
        assert 0.85 * model <= r['traffic'] <= 1.20 * model, (name, r['traffic'], model)
    assert line['scale_scene']['ms_per_step'] < 420.0 and line['n_gpus'] == 1 and line['config']['workload'].startswith('configs[2]')
