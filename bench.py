#!/usr/bin/env python
"""Headline benchmark (driver contract):  python bench.py --gpus N --steps K --warmup W

metric  : reconstructed points/sec (solve + mesh) at 1/2/4/8 MI355X; CG SpMV HBM GB/s        (BASELINE.json)
workload: at EVERY N the top-level ``value`` is BASELINE.json configs[4] -- the north_star scaling scene: synthetic 10M-point
          km-scale terrain (8 x 8 tiles of 125 m, tree_depth=5), recons_by_chunk over 64 chunks (examples/recons_by_chunk.py:26-29
          semantics: every chunk solved independently, blended), reconstruct(chunk_size=) + extract_dual_mesh(mise_iter=1).  The
          scene is the same at every N (STRONG scaling): with N ranks every rank generates only the tiles of its own chunks and
          their neighbours (sharded input), solves its chunks with no collective, exchanges chunk halos once (RCCL), meshes its
          cells; rank 0 gathers + stitches the mesh.  One "step" = one full pass of the hot path over the resident scene.
launch  : ``python bench.py --gpus N`` spawns its N ranks itself (torch.distributed.run, 127.0.0.1 rendezvous, one process per GPU,
          backend nccl = RCCL; NKSR_DIST_BACKEND=gloo lets the ranks share GPUs in tests) and fails loudly when fewer than N GPUs are
          visible; under an external ``torch.distributed.run`` (RANK / WORLD_SIZE in the environment) it joins that group instead.
roofline: the operator application inside the PCG loop of the timed region, measured live with HIP events on the solve stream.
          achieved = ALGORITHMIC bytes per application / average duration; ``achieved_physical`` counts the
          bytes the layout really moves.
N = 1 adds the sub-records the scene does not cover:
          cloud_1m          : configs[2] (synthetic 1M-point oriented cloud, detail_level=1.0, ONE field) with its operator roofline,
                              and the same workload through the assembled solve -> ``spmv_csr_roofline`` (north_star's SpMV KPI:
                              SURVEY.md section 8d, 8 nnz + 12 M + 4 bytes per launch, target 0.70 of 8 TB/s)
          small_inputs      : configs[0] / configs[1] latencies on the GPU (10 000-point bunny scan sequence, 3 000-point ShapeNet recipe)
          cpu_baseline      : the reference's examples/recons_waymo_cpu.py call sequence on assets/bunny.ply on the host cores
                              (oracle port, oracle/waymo_cpu.py), the GPU on the same input beside it
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver (RCCL peer mappings)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3e12 achievable)
TILE = 125.0       # metres, configs[4]: 8 x 8 chunks of 125 m
TILES = 8
PROFILES_DIR = os.path.join(ROOT, 'profiles')


# ---- committed PMC records -> roofline.traffic ----------------------------------------------------------------------------
def kernel_source_hash(kind):
    """Hash of the sources that define the roofline kernels (nksr_amd/build.py: kernel_hash) -- a committed PMC record is only
    attached to a bench line when it was taken on the same kernel code."""
    from nksr_amd import build
    return build.kernel_hash(kind)


def _load_pmc(suffix, match_key, value_key, bytes_per_launch, kind):
    """(bytes, source file, note).  A record qualifies when it was taken on the same system (its byte model within 2 % of this
    run's) AND on the same kernel sources (``kernel_source_hash``); the newest qualifying file wins.  A record of the same system
    from OTHER kernel sources is refused: traffic stays null and the note says which file was stale."""
    pdir = PROFILES_DIR
    best, stale = (None, None), None
    try:
        want = kernel_source_hash(kind)
    except Exception:
        want = None
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if not f.endswith(suffix):
                continue
            try:
                rec = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            if abs(rec.get(match_key, 0) - bytes_per_launch) >= 0.02 * bytes_per_launch:
                continue
            if want is not None and rec.get('kernel_source_hash') != want:
                stale = 'profiles/' + f
                continue
            best = (rec.get(value_key), 'profiles/' + f)
    note = None if best[0] is not None or stale is None else 'refused %s: taken on other kernel sources' % stale
    return best[0], best[1], note


def load_traffic(bytes_per_launch):
    """HBM bytes per SpMV launch from the committed rocprofv3 --pmc pass (profiles/*_spmv_pmc.json)."""
    return _load_pmc('_spmv_pmc.json', 'algorithmic_bytes_per_launch', 'hbm_bytes_per_launch', bytes_per_launch, 'spmv')


def load_traffic_fused(bytes_per_launch):
    """Same for the matrix-free operator (profiles/*_fused_pmc.json, matched on the operator's physical bytes)."""
    return _load_pmc('_fused_pmc.json', 'physical_bytes_per_application', 'hbm_bytes_per_application', bytes_per_launch, 'fused')


def live_traffic_fused(flags, timeout_s=240):
    """HBM bytes of one operator application collected IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- they do not
    fit one pass; --kernel-trace only, no other trace domain) over one untimed step of the same scene in child processes, after the
    timed region (nksr_amd/tools/scene_pmc.py, corrections as in MI355X_MICROARCH.md: KiB units, FETCH_SIZE doubled on gfx950).
    Returns (bytes or None, per-kernel record or None, note)."""
    import shutil
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, None, 'rocprofv3 not on PATH'
    fd, path = tempfile.mkstemp(suffix='.json', prefix='nksr_live_pmc_')
    os.close(fd)
    env = dict(os.environ, NKSR_PMC_GROUPS='2,3', NKSR_BENCH_CHILD='1', TMPDIR='/tmp')
    try:
        # (its own session: on a timeout the whole group goes -- scene_pmc -> rocprofv3 -> bench.py, not just the direct child, which
        # would leave a 10 M-point step running on the GPU under the measurements that follow)
        import signal
        proc = subprocess.Popen([sys.executable, '-m', 'nksr_amd.tools.scene_pmc', path] + list(flags), env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.communicate()
            raise

        class r:
            stderr = err
        rec = json.load(open(path))
        if 'hbm_bytes_per_application' not in rec:
            return None, None, 'counter passes gave no operator record: %s' % (r.stderr[-300:].replace('\n', ' '))
        return float(rec['hbm_bytes_per_application']), rec.get('operator_application'), None
    except Exception as e:      # (a profiler that fails must not fail the measurement)
        return None, None, 'live counter passes failed: %r' % (e,)
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def roofline_record(ms, launches, alg, phys, survey, fused=False, samples=None):
    """The CG operator application ("SpMV") measured live with HIP events on the solve stream inside the timed region.
    achieved / frac   = ALGORITHMIC bytes / time, a fraction of the 8 TB/s HBM peak (always <= 1):
                        assembled CSR: SURVEY.md section 8d, 8 nnz + 12 M + 4;
                        matrix-free operator: its algorithmic minimum -- 4 B per STORED entry of G and Q read
                        once + 4 B per row and level (row -> cell) + 108 B per unknown (one 27-entry stencil per cell) + x, y.
    achieved_physical = the bytes the implementation's layout really moves (packed 21-bit columns for the CSR; dense 27-slot rows
                        incl. structural zeros + partial blocks + tables for the operator) / time.
    survey_formula_*  = the same launches priced by SURVEY.md section 8d's matrix-free formula (2 x 8 B per stored entry + 12 M + 4):
                        the operator holds no column indices and reads every row once, so this exceeds what it moves -- kept for
                        comparison with round 2's line, NOT a utilisation figure.
    traffic           = HBM bytes per application from the committed rocprofv3 --pmc pass of the same system AND the same kernel
                        sources (static: counters are not collected in this run; ``traffic_source`` names the file, ``traffic_note``
                        says when a stale record was refused)."""
    avg_s = (ms / max(launches, 1)) * 1e-3
    a = alg / max(launches, 1)
    p = phys / max(launches, 1)
    sv = survey / max(launches, 1)
    rate = lambda b: (b / avg_s if avg_s > 0 else 0.0)
    traffic, src, note = ((load_traffic_fused(p) if fused else load_traffic(a)) if launches else (None, None, None))
    rec = {'bound': 'hbm',
           'kernel': ('k_fz_cells + k_fz_gather (+ k_fz_cellsum) (matrix-free normal-equation operator inside the PCG loop, fused_mode=True)' if fused else
                      'k_spmv<3,0> + k_spmv_fixup (packed-column CSR SpMV inside the PCG loop, fused_mode=False)'),
           'achieved': rate(a) / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': rate(a) / HBM_PEAK,
           'achieved_physical': rate(p) / 1e9, 'frac_physical': rate(p) / HBM_PEAK,
           'traffic': traffic, 'traffic_source': ('static: ' + src) if traffic is not None else None,
           'bytes_per_launch': a, 'physical_bytes_per_launch': p, 'avg_launch_us': avg_s * 1e6, 'launches_timed': launches}
    if note:
        rec['traffic_note'] = note
    if samples:
        # the launches one by one (same HIP events): a slow box or a few slow launches show in min / median, not in the mean
        us = sorted(1e3 * v for v in samples)
        med = us[len(us) // 2]
        rec.update(launch_us_min=us[0], launch_us_median=med, launch_us_max=us[-1],
                   frac_median=(a / (med * 1e-6) / HBM_PEAK if med > 0 else 0.0), frac_best=(a / (us[0] * 1e-6) / HBM_PEAK if us[0] > 0 else 0.0))
    if fused:
        rec['survey_formula_bytes_per_launch'] = sv
        rec['survey_formula_frac'] = rate(sv) / HBM_PEAK
    return rec


def mesh_adaptive_record(field, mise_iter, repeats=2):
    """The SAME solved field meshed on the adaptive dual graph (field.dual_graph = 'adaptive': cells as large as the level that
    carries them, models/nksr_net.py:132,214,284) and on the uniform lattice (the default), outside the timed region: milliseconds
    and triangle counts of both."""
    out = {}
    keep = getattr(field, 'dual_graph', 'lattice')
    try:
        for mode in ('lattice', 'adaptive'):
            field.dual_graph = mode
            m = field.extract_dual_mesh(mise_iter=mise_iter)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(repeats):
                m = field.extract_dual_mesh(mise_iter=mise_iter)
            torch.cuda.synchronize()
            out[mode] = {'ms': (time.perf_counter() - t0) / repeats * 1e3, 'vertices': int(m.v.shape[0]), 'triangles': int(m.f.shape[0])}
        out['adaptive_over_lattice'] = out['adaptive']['ms'] / max(out['lattice']['ms'], 1e-9)
    except Exception as e:      # (reported, never fatal: the headline is the lattice mesh)
        out['error'] = str(e)[:300]
    finally:
        field.dual_graph = keep
    return out


# ---- configs[4]: the scaling scene ------------------------------------------------------------------------------------
def terrain_setup(rec, dev, n_total, rank, world):
    """Sharded input of the 10M-point scene: the tiles of this rank's chunks and of their neighbours.  The chunk ->
    rank map is the Morton-contiguous partition of nksr_amd.dist on equal weights (every tile holds n_total / 64
    points), passed explicitly to reconstruct() so loader and solver agree."""
    from nksr_amd import dist as D, utils
    from nksr_amd.density import scale_for_detail_level
    per = n_total // (TILES * TILES)
    nchunk = TILES * TILES
    owner = D.partition_chunks(nchunk, world, [per] * nchunk, grid=(TILES, TILES, 1))
    t00 = utils.terrain_tile((0, 0), per, TILE, seed=0)[0]
    scale = scale_for_detail_level(torch.from_numpy(t00).to(dev), 1.0, rec.hparams.voxel_size)   # same tile on every rank
    need = set()
    for c in range(nchunk):
        if owner[c] == rank:
            cx, cy = c // TILES, c % TILES            # chunk id = (cx * grid_y + cy) * grid_z + cz
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    if 0 <= cx + dx < TILES and 0 <= cy + dy < TILES:
                        need.add((cx + dx, cy + dy))
    xs, ns = [], []
    for t in sorted(need):
        p, q = utils.terrain_tile(t, per, TILE, seed=0)
        xs.append(p)
        ns.append(q)
    xyz = torch.from_numpy(np.concatenate(xs) * np.float32(scale)).to(dev)
    nrm = torch.from_numpy(np.concatenate(ns)).to(dev)
    n_scene = per * nchunk
    bounds = ([0.0, 0.0, -40.0 * scale], [TILES * TILE * scale, TILES * TILE * scale, 40.0 * scale])
    return xyz, nrm, scale, owner, bounds, n_scene, len(need)


# ---- launch: one process per GPU ----------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, probe=False):
    """``python bench.py --gpus N`` without a process group: re-exec through torch.distributed.run, one rank per GPU.  Returns the
    exit code.  RCCL needs one device per rank: fewer than N visible GPUs is an error, not a silent 1-GPU run."""
    backend = os.environ.get('NKSR_DIST_BACKEND', 'nccl')
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend != 'gloo' and have < n:
        print('bench.py: --gpus %d needs %d visible MI355X devices for RCCL (one process per GPU); torch sees %d.  Nothing was run.  '
              '(NKSR_DIST_BACKEND=gloo lets the ranks share devices: a protocol test, not a measurement.)' % (n, n, have), file=sys.stderr)
        return 2
    if have < 1 and not (probe and backend == 'gloo'):
        print('bench.py: no GPU visible to PyTorch-ROCm', file=sys.stderr)
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, NKSR_BENCH_SPAWNED='1')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--points', type=int, default=1_000_000, help='configs[2] cloud size (cloud_1m sub-record)')
    ap.add_argument('--scene-points', type=int, default=10_000_000, help='configs[4] scene size (the headline)')
    ap.add_argument('--mise-iter', type=int, default=1)
    ap.add_argument('--detail-level', type=float, default=1.0)
    ap.add_argument('--cpu-sample', type=int, default=10000, help='points of the configs[2] crop every CPU worker solves')
    ap.add_argument('--cpu-cores', type=int, default=0, help='CPU baseline worker processes (0 = all host cores, at most 64)')
    ap.add_argument('--cpu-repeats', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--chunk-batch-points', type=int, default=0, help='chunk mode: points per batched solve (0 = the Reconstructor default, all chunks of a rank in one batch up to 2^25 points)')
    ap.add_argument('--no-cloud', action='store_true', help='skip the configs[2] sub-record (and with it spmv_csr_roofline)')
    ap.add_argument('--no-other-mode', action='store_true', help='skip the assembled-solve leg of the configs[2] sub-record')
    ap.add_argument('--no-small-inputs', action='store_true')
    ap.add_argument('--no-adaptive', action='store_true', help='skip the mesh_adaptive sub-records (the adaptive dual graph timed beside the lattice mesher, outside the timed region)')
    ap.add_argument('--dual-graph', choices=['lattice', 'adaptive'], default='lattice', help="the mesher of the headline scene: the uniform lattice (default; the "
                    "metric is quoted on it) or the adaptive dual graph (Reconstructor.dual_graph: deeper halos between ranks, cell-pair vertex names)")
    ap.add_argument('--no-live-pmc', action='store_true', help='do not collect roofline.traffic with rocprofv3 in this run (two counter passes over one extra step)')
    ap.add_argument('--cloud-steps', type=int, default=3)
    ap.add_argument('--dist-probe', action='store_true', help='launch check only: start the N ranks, run the handshake collectives over the '
                                                              'backend, print the ``dist`` record and exit (no reconstruction)')
    args = ap.parse_args()

    if args.gpus < 1:
        print('bench.py: --gpus must be >= 1', file=sys.stderr)
        return 2
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:], probe=args.dist_probe)

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        print('bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): they must agree' % (args.gpus, world), file=sys.stderr)
        return 2
    dist = None
    backend = os.environ.get('NKSR_DIST_BACKEND', 'nccl')
    ndev = max(torch.cuda.device_count(), 1)
    if args.dist_probe:
        return dist_probe(rank, local_rank, world, backend)
    if world > 1 and backend != 'gloo' and ndev < world:
        print('bench.py: %d ranks but %d visible GPUs: RCCL needs one device per rank' % (world, ndev), file=sys.stderr)
        return 2
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)            # before the process group: RCCL binds to the current device
    dev = torch.device('cuda', local_rank)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # backend "nccl" is RCCL on ROCm; NKSR_DIST_BACKEND=gloo lets two ranks share one GPU in tests
        dist.init_process_group(backend, rank=rank, world_size=world)
        # create the communicator (RCCL ring / peer mappings) now, outside every timed region; the sum of ones over the backend
        # is the number of ranks that really took part
        warm = torch.ones(1, device=dev if dist.get_backend() != 'gloo' else 'cpu')
        dist.all_reduce(warm)
        ranks_seen = int(round(float(warm.item())))
        dist.all_gather([torch.zeros_like(warm) for _ in range(world)], warm)

    import nksr_amd
    from nksr_amd import configs, solver, utils

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_loop(step, steps, warmup, stage_acc):
        """W untimed warm-up steps, then EXACTLY K timed steps between two barrier + device-sync fences; max over ranks."""
        for _ in range(warmup):
            step(None)
        stage_acc.clear()
        solver.profile_spmv(True)
        solver.profile_spmv_bytes()
        fence()
        ms0 = torch.cuda.memory_stats(dev)
        t0 = time.perf_counter()
        out = None
        diag = os.environ.get('NKSR_BENCH_STEP_TIMES', '') == '1'      # diagnostics only: a device sync after every step
        for _ in range(steps):
            # the previous result is released first: every step then has the memory footprint of the warm-up step and
            # the caching allocator serves it from its pool
            out = None
            ts = time.perf_counter()
            out = step(stage_acc)
            if diag:
                torch.cuda.synchronize()
                m = torch.cuda.memory_stats(dev)
                print('[step] %.1f ms, device allocs so far %d, frees %d, reserved %.1f GB, peak allocated %.1f GB' % (
                    (time.perf_counter() - ts) * 1e3, m.get('num_device_alloc', 0), m.get('num_device_free', 0),
                    m.get('reserved_bytes.all.current', 0) / 1e9, m.get('allocated_bytes.all.peak', 0) / 1e9), file=sys.stderr)
        fence()
        dt = time.perf_counter() - t0
        ms1 = torch.cuda.memory_stats(dev)
        # hipMalloc / hipFree calls inside the timed region (0 in the steady state: the caching allocator serves every step from its pool)
        alloc = {'device_allocs_per_step': (ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)) / steps,
                 'device_frees_per_step': (ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0)) / steps,
                 'reserved_GB': ms1.get('reserved_bytes.all.current', 0) / 1e9, 'peak_allocated_GB': ms1.get('allocated_bytes.all.peak', 0) / 1e9}
        ms, launches = solver.profile_spmv(False)
        samples = solver.profile_spmv_samples()
        alg, phys, survey = solver.profile_spmv_bytes()
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() != 'gloo' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, out, (ms, launches, alg, phys, survey, samples), alloc

    def acc_stages(acc, timing, extra):
        if acc is not None:
            for k, v in list(timing.items()) + list(extra.items()):
                acc[k] = acc.get(k, 0.0) + v

    # ---- configs[4]: strong-scaling scene ----------------------------------------------------------------------------
    def run_terrain(steps, warmup, fused):
        rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
        rec.sync_timing = True
        rec.dual_graph = args.dual_graph
        if args.chunk_batch_points > 0:
            rec.chunk_batch_points = args.chunk_batch_points
        xyz, nrm, scale, owner, bounds, n_scene, ntiles = terrain_setup(rec, dev, args.scene_points, rank, world)
        chunk_size = TILE * scale

        def step(acc):
            field = rec.reconstruct(xyz, nrm, detail_level=None, chunk_size=chunk_size, sharded_input=True, chunk_owner=owner,
                                    chunk_bounds=bounds, fused_mode=fused)
            t0 = time.perf_counter()
            mesh = field.extract_dual_mesh(mise_iter=args.mise_iter)
            torch.cuda.synchronize()
            acc_stages(acc, rec.timing, {'t_mesh': time.perf_counter() - t0, 't_gather': getattr(field, 'last_gather_s', 0.0)})
            return field, mesh

        acc = {}
        dt, (field, mesh), prof, alloc = timed_loop(step, steps, warmup, acc)
        adaptive = mesh_adaptive_record(field, args.mise_iter) if (world == 1 and not args.no_adaptive) else None
        infos = field.chunk_infos()
        cfg = {'workload': 'configs[4]: synthetic %d-point km-scale terrain + boxes (8x8 tiles of 125 m), tree_depth=5, chunk_size=125 m '
                           '(64 chunks), reconstruct(chunk_size=)+extract_dual_mesh(mise_iter=%d), the same scene at every N (strong scaling)' % (n_scene, args.mise_iter),
               'scene_points': n_scene, 'fused_mode': fused, 'tree_depth': 5, 'kernel_dim': rec.hparams.kernel_dim, 'global_scale': scale,
               'chunks': TILES * TILES, 'batched_solves_this_rank': len([p for p in field.parts if p.solved]), 'chunks_this_rank': sum(1 for c in owner if c == rank), 'tiles_loaded_this_rank': ntiles,
               'points_resident_this_rank': int(xyz.shape[0]),
               'unknowns_M_per_chunk': int(np.mean([i['M'] for i in infos])) if infos else 0,
               'pcg_iters_per_chunk': float(np.mean([i['iters'] for i in infos])) if infos else 0,
               'pcg_iters_max_chunk': int(max([i['iters'] for i in infos])) if infos else 0,
               'pcg_iters_min_chunk': int(min([i['iters'] for i in infos])) if infos else 0,
               'jacobi_fallbacks': int(sum(int(getattr(p.field, 'solve_info', {}).get('jacobi_fallbacks', 0) or 0) for p in field.parts if p.solved)),
               'mesh_vertices': int(mesh.v.shape[0]), 'mesh_triangles': int(mesh.f.shape[0]),
               'dual_graph': args.dual_graph,
               'parallelism': 'none' if world == 1 else 'chunks sharded over %d ranks (Morton-contiguous), sharded input, no collective in the solve, '
                                                        'one halo exchange, mesh gather + stitch on rank 0' % world}
        from nksr_amd.fields import kernel_field as _kf
        if _kf.DETAIL_TIMES:
            cfg['detail_ms_total'] = {k: round(v * 1e3, 1) for k, v in sorted(_kf.DETAIL_TIMES.items()) if k != '_'}
        if adaptive is not None:
            cfg['mesh_adaptive'] = adaptive
        return dt, n_scene, cfg, prof, {k: v / steps for k, v in sorted(acc.items())}, alloc

    # ---- configs[2]: the single-field roofline workload -----------------------------------------------------------------
    def run_cloud(steps, warmup, fused):
        rec = nksr_amd.Reconstructor(dev)
        rec.sync_timing = True
        xyz_np, nrm_np = utils.synth_scene(args.points, seed=0, extent=(40.0, 40.0, 10.0), noise=0.01)
        xyz, nrm = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(nrm_np).to(dev)

        def step(acc):
            field = rec.reconstruct(xyz, nrm, detail_level=args.detail_level, fused_mode=fused)
            t0 = time.perf_counter()
            mesh = field.extract_dual_mesh(mise_iter=args.mise_iter)
            torch.cuda.synchronize()
            acc_stages(acc, rec.timing, {'t_mesh': time.perf_counter() - t0})
            return field, mesh

        acc = {}
        dt, (field, mesh), prof, alloc = timed_loop(step, steps, warmup, acc)
        adaptive = mesh_adaptive_record(field, args.mise_iter) if (fused and not args.no_adaptive) else None
        info = field.solve_info
        cfg = {'workload': 'configs[2]: synthetic %d-point oriented cloud (8 spheres/tori in a 40x40x10 box, sigma=0.01), '
                           'detail_level=%.1f, reconstruct()+extract_dual_mesh(mise_iter=%d)' % (args.points, args.detail_level, args.mise_iter),
               'points': args.points, 'fused_mode': fused, 'tree_depth': rec.hparams.tree_depth, 'kernel_dim': rec.hparams.kernel_dim,
               'unknowns_M': info['M'], 'nnz_A': info['nnz'], 'kernel_row_slots': info.get('kernel_row_slots'),
               'stored_entries_G_Q': field.stored_entries() if fused else None, 'pcg_iters': info['iters'], 'pcg_rel_residual': info['rel_residual'],
               'mesh_vertices': int(mesh.v.shape[0]), 'mesh_triangles': int(mesh.f.shape[0]), 'global_scale': field.scale}
        if adaptive is not None:
            cfg['mesh_adaptive'] = adaptive
        return dt, args.points, cfg, prof, {k: v / steps for k, v in sorted(acc.items())}, alloc, (rec, xyz_np, nrm_np, field.scale)

    # The headline runs through the API default, fused_mode=True (what the reference's examples pass): chunk mode = the batched
    # matrix-free solve.
    dt, npts, cfg, prof, stages, alloc = run_terrain(args.steps, args.warmup, True)
    out = {
        'metric': 'reconstructed points/sec (solve+mesh)', 'value': npts * args.steps / dt, 'unit': 'points/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic', 'headline_config': 'configs[4]',      # (the top-level workload since round 4; rounds 1-3 quoted configs[2], now the cloud_1m sub-record)
        'config': cfg, 'roofline': roofline_record(*prof[:5], fused=True, samples=prof[5]), 'stages_s_per_step': stages, 'allocator': alloc,
        'dist': {'backend': (dist.get_backend() if dist is not None else None), 'world_size': world, 'rccl_ranks_seen': ranks_seen,
                 'launcher': 'bench.py self-spawn (torch.distributed.run)' if os.environ.get('NKSR_BENCH_SPAWNED') else ('external' if world > 1 else 'single process')},
    }
    if dist is not None:
        # per-rank stage times (seconds per step): what every rank spent where, incl. the halo exchange and the mesh gather
        mine = {'rank': rank, 'chunks': cfg['chunks_this_rank'], 'points_resident': cfg['points_resident_this_rank']}
        mine.update({k: round(v, 5) for k, v in stages.items()})
        per = [None] * world
        dist.all_gather_object(per, mine)
        out['dist']['per_rank'] = per
    if world == 1 and not args.no_cloud:
        torch.cuda.empty_cache()
        cdt, cn, ccfg, cprof, cstages, calloc, extra = run_cloud(args.cloud_steps, 1, True)
        cloud = {'value': cn * args.cloud_steps / cdt, 'unit': 'points/s', 'ms_per_step': cdt / args.cloud_steps * 1e3, 'steps': args.cloud_steps, 'warmup': 1,
                 'config': ccfg, 'roofline': roofline_record(*cprof[:5], fused=True, samples=cprof[5]), 'stages_s_per_step': cstages, 'allocator': calloc}
        if not args.no_other_mode:
            # the same workload through the other solve: assembled CSR + streaming SpMV (solve_non_fused, the path training needs)
            # (five steps: >= 50 timed SpMV launches)
            OSTEPS = 5
            odt, _, ocfg, oprof, ostages, oalloc, _ = run_cloud(OSTEPS, 1, False)
            cloud['other_solve_mode'] = {'fused_mode': False, 'value': cn * OSTEPS / odt, 'unit': 'points/s', 'ms_per_step': odt / OSTEPS * 1e3, 'steps': OSTEPS,
                                         'warmup': 1, 'unknowns_M': ocfg['unknowns_M'], 'nnz_A': ocfg['nnz_A'], 'pcg_iters': ocfg['pcg_iters'],
                                         'roofline': roofline_record(*oprof[:5], fused=False, samples=oprof[5]), 'stages_s_per_step': ostages, 'allocator': oalloc}
            # north_star's KPI at the top level of every N = 1 line: the CSR SpMV roofline (target 0.70 of 8 TB/s)
            out['spmv_csr_roofline'] = cloud['other_solve_mode']['roofline']
        out['cloud_1m'] = cloud
    else:
        extra = None
    if world == 1 and not args.no_small_inputs:
        out['small_inputs'] = small_inputs(dev)
    if rank == 0 and world == 1 and not args.no_live_pmc and not os.environ.get('NKSR_BENCH_CHILD') and out.get('roofline'):
        # same-run counter figure for roofline.traffic (the committed record of profiles/ stays the fallback)
        torch.cuda.empty_cache()
        tb, per_kernel, note = live_traffic_fused(['--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-cloud', '--no-small-inputs', '--no-live-pmc',
                                                   '--scene-points', str(args.scene_points), '--mise-iter', str(args.mise_iter),
                                                   '--chunk-batch-points', str(args.chunk_batch_points)])
        if tb is not None:
            out['roofline']['traffic'] = tb
            out['roofline']['traffic_source'] = 'live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over one step of this run (nksr_amd/tools/scene_pmc.py)'
            out['roofline']['traffic_per_kernel'] = per_kernel
            out['roofline'].pop('traffic_note', None)
        elif note:
            out['roofline']['traffic_live_note'] = note
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(dev, args, extra)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def dist_probe(rank, local_rank, world, backend):
    """The launch path without the workload: every rank joins the group, the handshake collectives of the real run go over the
    backend (all_reduce of ones = ranks seen, all_gather, a point-to-point ring), rank 0 prints the record."""
    import torch.distributed as dist
    on_gpu = backend != 'gloo' and torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    rec = {'dist_probe': True, 'n_gpus': world, 'backend': backend, 'rccl_ranks_seen': 1, 'launcher': 'bench.py self-spawn (torch.distributed.run)' if os.environ.get('NKSR_BENCH_SPAWNED') else 'external'}
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rec['rccl_ranks_seen'] = int(round(float(one.item())))
        got = [torch.zeros(1, device=dev) for _ in range(world)]
        dist.all_gather(got, torch.full((1,), float(rank), device=dev))
        rec['all_gather_ranks'] = [int(g.item()) for g in got]
        # point-to-point ring (the mesh gather's transport): rank r sends r to r + 1
        recv = torch.full((1,), -1.0, device=dev)
        ops = [dist.P2POp(dist.isend, torch.full((1,), float(rank), device=dev), (rank + 1) % world),
               dist.P2POp(dist.irecv, recv, (rank - 1) % world)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        ok = torch.tensor([1.0 if int(recv.item()) == (rank - 1) % world else 0.0], device=dev)
        dist.all_reduce(ok)
        rec['p2p_ring_ok'] = int(round(float(ok.item()))) == world
        dist.barrier()
    if rank == 0:
        print(json.dumps(rec))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    return 0


def _latency(fn, repeats):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / repeats * 1e3, r


def small_inputs(dev):
    """configs[1] on the GPU: the ShapeNet-3K-noise recipe (3 000 points, sigma 0.005, preset snet-n3k-wnormal) end to end --
    launch-latency bound, reported as a latency."""
    import nksr_amd
    from nksr_amd import utils
    xyz, nrm = utils.synth_sphere(3000, 0.45, 0.005, 0)
    x, n = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev, config='snet-n3k-wnormal')
    st = {}

    def seq():
        f = rec.reconstruct(x, n, detail_level=None)
        st['iters'] = f.solve_info['iters']
        return f.extract_dual_mesh(mise_iter=1)
    ms, m = _latency(seq, 10)
    return {'configs1_shapenet_3k': {'ms': ms, 'points': 3000, 'pcg_iters': st['iters'], 'mesh_triangles': int(m.f.shape[0]),
                                     'sample': 'sphere r=0.45, N=3000, sigma=0.005, preset snet-n3k-wnormal, reconstruct(detail_level=None)+extract_dual_mesh(mise_iter=1), mean of 10'}}


def cpu_baseline(dev, args, extra):
    import nksr_amd
    from oracle import waymo_cpu
    crop = None
    if extra is not None:
        rec, xyz_np, nrm_np, scale = extra
        # bounded crop of the configs[2] cloud (model units), one per CPU worker
        c = xyz_np[0]
        idx = np.argsort(np.abs(xyz_np - c).max(1))[:args.cpu_sample]
        crop = os.path.join(tempfile.gettempdir(), 'nksr_bench_crop_%d.npz' % os.getpid())
        np.savez(crop, xyz=(xyz_np[idx] * np.float32(scale)).astype(np.float32), normal=nrm_np[idx], mise_iter=args.mise_iter)
    else:
        rec = nksr_amd.Reconstructor(dev)
    try:
        cb = waymo_cpu.measure(cores=args.cpu_cores or min(os.cpu_count() or 1, 64), repeats=args.cpu_repeats, crop=crop)
    finally:
        if crop and os.path.exists(crop):
            os.remove(crop)
    # the GPU on the same input (same call sequence, sensor-only, through the product's preprocess_fn)
    d = np.load(waymo_cpu.BUNNY)
    bx = torch.from_numpy(d['xyz']).to(dev)
    bs = torch.from_numpy(waymo_cpu.synth_sensors(d['xyz'], d['normal'])).to(dev)
    fn = nksr_amd.get_estimate_normal_preprocess_fn(64, 85.0)

    def gpu_seq():
        f = rec.reconstruct(bx, sensor=bs, detail_level=None, approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True, preprocess_fn=fn)
        return f.extract_dual_mesh(mise_iter=1)
    ms, m = _latency(gpu_seq, 5)
    cb['gpu_same_input'] = {'value': bx.shape[0] / (ms * 1e-3), 'unit': 'points/s', 'ms': ms, 'mesh_triangles': int(m.f.shape[0]),
                            'note': 'one 10 000-point scan at a time on one MI355X: launch-latency bound, not a throughput figure'}
    return cb


if __name__ == '__main__':
    sys.exit(main())
