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
    assert (set(r) - {'traffic_note'}) | {'survey_formula_bytes_per_launch', 'survey_formula_frac'} == set(f) - {'traffic_note'} and 'k_fz_cells' in f['kernel']
    assert abs(f['achieved'] - 1.60e9 / 0.56e-3 / 1e9) < 1e-6 * f['achieved'] and abs(f['achieved_physical'] - 2.58e9 / 0.56e-3 / 1e9) < 1e-6 * f['achieved']
    assert 0 < f['frac'] < f['frac_physical'] <= 1.0 < f['survey_formula_frac']
    z = b.roofline_record(0.0, 0, 0.0, 0.0, 0.0)
    assert z['achieved'] == 0.0 and z['traffic'] is None


def _run_bench(args, env=None, timeout=300):
    import subprocess
    import sys
    e = dict(os.environ)
    e.pop('WORLD_SIZE', None)
    e.pop('RANK', None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_n_without_n_devices_fails_loudly():
    """``python bench.py --gpus N`` (the driver's command, no torchrun around it) must never degrade to a 1-GPU run: with fewer than
    N visible devices it runs nothing, says why, and exits non-zero."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('needs a box with fewer than 2 GPUs')
    r = _run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0'], env={'NKSR_DIST_BACKEND': 'nccl'})
    assert r.returncode == 2 and 'needs 2 visible' in r.stderr and 'Nothing was run' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    # a launcher that started another number of ranks than --gpus says is refused too
    r = _run_bench(['--gpus', '4', '--dist-probe'], env={'WORLD_SIZE': '2', 'RANK': '0'})
    assert r.returncode == 2 and 'must agree' in r.stderr


def test_gpus_n_spawns_its_own_ranks():
    """The self-spawn path end to end on CPU (gloo): ``--gpus 2 --dist-probe`` starts two ranks through torch.distributed.run on
    127.0.0.1, the handshake collectives of the real run (all_reduce of ones = ranks seen, all_gather, a point-to-point ring) go
    over the backend, rank 0 prints one record with n_gpus = 2."""
    import json
    r = _run_bench(['--gpus', '2', '--dist-probe'], env={'NKSR_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks_seen'] == 2 and d['all_gather_ranks'] == [0, 1] and d['p2p_ring_ok'] is True
    assert 'self-spawn' in d['launcher'] and d['backend'] == 'gloo'


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


def test_stale_pmc_records_are_refused(tmp_path):
    """``roofline.traffic`` comes from a committed rocprofv3 --pmc record only when that record was taken on the same system (byte
    model within 2 %) AND on the same kernel sources (nksr_amd/build.py: kernel_hash): a record from older kernels is refused,
    traffic stays null and the line says which file was stale."""
    import json
    from nksr_amd import build
    b = _bench()
    b.PROFILES_DIR = str(tmp_path)
    phys = 2.58e9
    json.dump({'physical_bytes_per_application': phys, 'hbm_bytes_per_application': 2.57e9, 'kernel_source_hash': 'deadbeef'},
              open(os.path.join(str(tmp_path), 'r00_fused_pmc.json'), 'w'))
    got, src, note = b.load_traffic_fused(phys)
    assert got is None and src is None and 'r00_fused_pmc.json' in note and 'other kernel sources' in note
    r = b.roofline_record(160 * 0.56, 160, 160 * 1.60e9, 160 * phys, 160 * 5.65e9, fused=True)
    assert r['traffic'] is None and r['traffic_source'] is None and 'refused' in r['traffic_note']
    json.dump({'physical_bytes_per_application': phys, 'hbm_bytes_per_application': 2.57e9, 'kernel_source_hash': build.kernel_hash('fused')},
              open(os.path.join(str(tmp_path), 'r01_fused_pmc.json'), 'w'))
    got, src, note = b.load_traffic_fused(phys)
    assert got == 2.57e9 and src == 'profiles/r01_fused_pmc.json' and note is None
    # another system (other byte model): not attached, nothing to refuse
    assert b.load_traffic_fused(2.0 * phys) == (None, None, None)
    assert build.kernel_hash('fused') != build.kernel_hash('spmv') and len(build.kernel_hash('spmv')) == 16


def test_committed_bench_line_and_its_pmc_records():
    """The round's committed bench line (profiles/r05_bench.json): configs[4] is the top-level workload at N = 1, the configs[2]
    single-field workload and north_star's CSR SpMV KPI travel as sub-records, every roofline fraction is a fraction, and wherever
    a line carries ``traffic`` the committed PMC record it names exists, carries a kernel source hash and agrees with the byte
    model the physical fraction is priced on (no hidden re-reads)."""
    import json
    import pytest
    path = os.path.join(ROOT, 'profiles', 'r05_bench.json')
    if not os.path.exists(path):
        pytest.skip('profiles/r05_bench.json is committed with the round\'s GPU run')
    line = json.loads(open(path).read().strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['config']['workload'].startswith('configs[4]') and line['scaling'] == 'strong'
    assert line['cloud_1m']['config']['workload'].startswith('configs[2]') and line['dist']['rccl_ranks_seen'] == 1
    recs = [('scene', line['roofline']), ('cloud', line['cloud_1m']['roofline']), ('csr', line['spmv_csr_roofline'])]
    for name, r in recs:
        assert 0.0 < r['frac'] <= 1.0 and r['frac'] <= r['frac_physical'] * 1.3, name
        assert (r['traffic'] is None) == (r['traffic_source'] is None), name
        if r['traffic'] is not None:
            if r['traffic_source'].startswith('static: '):      # a committed counter record; 'live: ...' = collected in the run itself (round 5)
                src = os.path.join(ROOT, r['traffic_source'][len('static: '):])
                assert os.path.exists(src), src
                assert len(json.load(open(src)).get('kernel_source_hash', '')) == 16, src
            else:
                assert r['traffic_source'].startswith('live: ')
            model = r['physical_bytes_per_launch']
            assert 0.85 * model <= r['traffic'] <= 1.20 * model, (name, r['traffic'], model)
    assert line['ms_per_step'] < 420.0 and 'device_allocs_per_step' in line['allocator']
