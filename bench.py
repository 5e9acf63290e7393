#!/usr/bin/env python
"""Headline benchmark (driver contract):  python bench.py --gpus N --steps K --warmup W

metric  : reconstructed points/sec (solve + mesh); CG SpMV HBM GB/s        (BASELINE.json)
N = 1   : BASELINE.json configs[2] -- synthetic 1M-point oriented cloud, detail_level=1.0, reconstruct() +
          extract_dual_mesh(mise_iter=1): the configuration the SpMV roofline is quoted on.  One "step" = one full pass
          of the hot path over the resident cloud.  The line also carries
            scale_scene  : configs[4] (below) on this one GPU, warm -- the N=1 point of the scaling curve
            cpu_baseline : the reference's examples/recons_waymo_cpu.py call sequence on assets/bunny.ply on the host
                           cores (oracle port, oracle/waymo_cpu.py), the GPU on the same input beside it, and a bounded
                           crop of the configs[2] cloud through the oracle
N > 1   : BASELINE.json configs[4] -- the north_star scaling scene: synthetic 10M-point km-scale terrain (8 x 8 tiles of
          125 m, tree_depth=5), recons_by_chunk over 64 chunks, STRONG scaling: the same scene on 1/2/4/8 ranks.  Every
          rank generates only the tiles of its own chunks and their neighbours (sharded input), solves its chunks with no
          collective, exchanges chunk halos once (RCCL), meshes its cells; rank 0 gathers + stitches the mesh.
roofline: the operator application inside the PCG loop.  achieved = ALGORITHMIC bytes per application (SURVEY.md section 8d:
          assembled CSR 8 nnz + 12 M + 4; matrix-free 2 x 8 bytes per stored entry of G and Q + 12 M + 4) / average duration,
          measured live with HIP events on the solve stream inside the timed region; ``achieved_physical`` counts the bytes
          the layout really moves (packed columns / index-free rows read once).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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


def load_traffic(bytes_per_launch):
    """HBM bytes per SpMV launch from the committed rocprofv3 --pmc pass (profiles/*_spmv_pmc.json), used only when
    that pass was taken on the same matrix as this run (same algorithmic bytes).  Returns (bytes, source file)."""
    best = (None, None)
    pdir = os.path.join(ROOT, 'profiles')
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith('_spmv_pmc.json'):
                try:
                    rec = json.load(open(os.path.join(pdir, f)))
                    if abs(rec.get('algorithmic_bytes_per_launch', 0) - bytes_per_launch) < 0.02 * bytes_per_launch:
                        best = (rec.get('hbm_bytes_per_launch'), 'profiles/' + f)
                except Exception:
                    pass
    return best


def load_traffic_fused(bytes_per_launch):
    """Same for the matrix-free operator (profiles/*_fused_pmc.json, matched on the operator's physical bytes)."""
    best = (None, None)
    pdir = os.path.join(ROOT, 'profiles')
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith('_fused_pmc.json'):
                try:
                    rec = json.load(open(os.path.join(pdir, f)))
                    if abs(rec.get('physical_bytes_per_application', 0) - bytes_per_launch) < 0.02 * bytes_per_launch:
                        best = (rec.get('hbm_bytes_per_application'), 'profiles/' + f)
                except Exception:
                    pass
    return best


def roofline_record(ms, launches, alg, phys, survey, fused=False):
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
    traffic           = HBM bytes per application from the committed rocprofv3 --pmc pass of the same system (static: not measured in
                        this run; ``traffic_source`` names the file)."""
    avg_s = (ms / max(launches, 1)) * 1e-3
    a = alg / max(launches, 1)
    p = phys / max(launches, 1)
    sv = survey / max(launches, 1)
    rate = lambda b: (b / avg_s if avg_s > 0 else 0.0)
    traffic, src = ((load_traffic_fused(p) if fused else load_traffic(a)) if launches else (None, None))
    rec = {'bound': 'hbm',
           'kernel': ('k_fz_sweep + k_fz_cellsum + k_fz_gather (matrix-free normal-equation operator inside the PCG loop, fused_mode=True)' if fused else
                      'k_spmv<3,0> + k_spmv_fixup (packed-column CSR SpMV inside the PCG loop, fused_mode=False)'),
           'achieved': rate(a) / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': rate(a) / HBM_PEAK,
           'achieved_physical': rate(p) / 1e9, 'frac_physical': rate(p) / HBM_PEAK,
           'traffic': traffic, 'traffic_source': ('static: ' + src) if traffic is not None else None,
           'bytes_per_launch': a, 'physical_bytes_per_launch': p, 'avg_launch_us': avg_s * 1e6, 'launches_timed': launches}
    if fused:
        rec['survey_formula_bytes_per_launch'] = sv
        rec['survey_formula_frac'] = rate(sv) / HBM_PEAK
    return rec


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--points', type=int, default=1_000_000, help='configs[2] cloud size (N=1 headline)')
    ap.add_argument('--scene-points', type=int, default=10_000_000, help='configs[4] scene size')
    ap.add_argument('--mise-iter', type=int, default=1)
    ap.add_argument('--detail-level', type=float, default=1.0)
    ap.add_argument('--cpu-sample', type=int, default=10000, help='points of the configs[2] crop every CPU worker solves')
    ap.add_argument('--cpu-cores', type=int, default=0, help='CPU baseline worker processes (0 = all host cores, at most 64)')
    ap.add_argument('--cpu-repeats', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--chunk-batch-points', type=int, default=0, help='chunk mode: points per batched solve (0 = the Reconstructor default, all chunks of a rank in one batch up to 2^25 points)')
    ap.add_argument('--no-scale-scene', action='store_true')
    ap.add_argument('--no-other-mode', action='store_true')
    ap.add_argument('--non-fused', action='store_true', help='configs[2] headline through the assembled CSR solve (fused_mode=False)')
    ap.add_argument('--scene', choices=['auto', 'cloud', 'terrain'], default='auto',
                    help="'terrain' runs configs[4] as the headline at N=1 too")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)            # before the process group: RCCL binds to the current device
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # backend "nccl" is RCCL on ROCm; NKSR_DIST_BACKEND=gloo lets two ranks share one GPU in tests
        dist.init_process_group(os.environ.get('NKSR_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
        # create the communicator (RCCL ring / peer mappings) now, outside every timed region
        warm = torch.zeros(1, device=dev if dist.get_backend() != 'gloo' else 'cpu')
        dist.all_reduce(warm)
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
        stage_acc['device_allocs'] = float(ms1.get('num_device_alloc', 0) - ms0.get('num_device_alloc', 0)) * steps
        stage_acc['device_frees'] = float(ms1.get('num_device_free', 0) - ms0.get('num_device_free', 0)) * steps
        ms, launches = solver.profile_spmv(False)
        alg, phys, survey = solver.profile_spmv_bytes()
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, out, (ms, launches, alg, phys, survey)

    def acc_stages(acc, rec, tm):
        if acc is not None:
            for k, v in list(rec.timing.items()) + [('t_mesh', tm)]:
                acc[k] = acc.get(k, 0.0) + v

    # ---- configs[4]: strong-scaling scene ----------------------------------------------------------------------------
    def run_terrain(steps, warmup, fused):
        rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
        rec.sync_timing = True
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
            acc_stages(acc, rec, time.perf_counter() - t0)
            return field, mesh

        acc = {}
        dt, (field, mesh), prof = timed_loop(step, steps, warmup, acc)
        infos = field.chunk_infos()
        cfg = {'workload': 'configs[4]: synthetic %d-point km-scale terrain + boxes (8x8 tiles of 125 m), tree_depth=5, chunk_size=125 m '
                           '(64 chunks), reconstruct(chunk_size=)+extract_dual_mesh(mise_iter=%d), STRONG scaling' % (n_scene, args.mise_iter),
               'scene_points': n_scene, 'fused_mode': fused, 'tree_depth': 5, 'kernel_dim': rec.hparams.kernel_dim, 'global_scale': scale,
               'chunks': TILES * TILES, 'batched_solves_this_rank': len([p for p in field.parts if p.solved]), 'chunks_this_rank': sum(1 for c in owner if c == rank), 'tiles_loaded_this_rank': ntiles,
               'points_resident_this_rank': int(xyz.shape[0]),
               'unknowns_M_per_chunk': int(np.mean([i['M'] for i in infos])) if infos else 0,
               'pcg_iters_per_chunk': float(np.mean([i['iters'] for i in infos])) if infos else 0,
               'pcg_iters_max_chunk': int(max([i['iters'] for i in infos])) if infos else 0,
               'pcg_iters_min_chunk': int(min([i['iters'] for i in infos])) if infos else 0,
               'mesh_vertices': int(mesh.v.shape[0]), 'mesh_triangles': int(mesh.f.shape[0]),
               'parallelism': 'none' if world == 1 else 'chunks sharded over %d ranks (Morton-contiguous), sharded input, no collective in the solve, '
                                                        'one halo exchange, mesh gather + stitch on rank 0' % world}
        from nksr_amd.fields import kernel_field as _kf
        if _kf.DETAIL_TIMES:
            cfg['detail_ms_total'] = {k: round(v * 1e3, 1) for k, v in sorted(_kf.DETAIL_TIMES.items()) if k != '_'}
        return dt, n_scene, cfg, prof, {k: v / steps for k, v in sorted(acc.items())}

    # ---- configs[2]: the roofline workload ---------------------------------------------------------------------------
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
            acc_stages(acc, rec, time.perf_counter() - t0)
            return field, mesh

        acc = {}
        dt, (field, mesh), prof = timed_loop(step, steps, warmup, acc)
        info = field.solve_info
        cfg = {'workload': 'configs[2]: synthetic %d-point oriented cloud (8 spheres/tori in a 40x40x10 box, sigma=0.01), '
                           'detail_level=%.1f, reconstruct()+extract_dual_mesh(mise_iter=%d)' % (args.points, args.detail_level, args.mise_iter),
               'points': args.points, 'fused_mode': fused, 'tree_depth': rec.hparams.tree_depth, 'kernel_dim': rec.hparams.kernel_dim,
               'unknowns_M': info['M'], 'nnz_A': info['nnz'], 'kernel_row_slots': info.get('kernel_row_slots'),
               'stored_entries_G_Q': field.stored_entries() if fused else None, 'pcg_iters': info['iters'], 'pcg_rel_residual': info['rel_residual'],
               'mesh_vertices': int(mesh.v.shape[0]), 'mesh_triangles': int(mesh.f.shape[0]), 'global_scale': field.scale,
               'parallelism': 'none'}
        return dt, args.points, cfg, prof, {k: v / steps for k, v in sorted(acc.items())}, (rec, xyz_np, nrm_np, field.scale)

    terrain_headline = world > 1 or args.scene == 'terrain'
    extra = None
    # Both workloads run through the API default, fused_mode=True (what the reference's examples pass): the matrix-free solve
    # skips the assembly and its operator (one pass over the index-free kernel rows) is about as fast per application as the CSR
    # SpMV; the other mode is measured and reported next to each (DESIGN.md section 3.5 has the cost model).
    fused = (not args.non_fused) if not terrain_headline else True      # chunk mode = the batched matrix-free solve
    if terrain_headline:
        dt, npts, cfg, prof, stages = run_terrain(args.steps, args.warmup, fused)
    else:
        dt, npts, cfg, prof, stages, extra = run_cloud(args.steps, args.warmup, fused)
    out = {
        'metric': 'reconstructed points/sec (solve+mesh)', 'value': npts * args.steps / dt, 'unit': 'points/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong' if terrain_headline else 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic', 'config': cfg, 'roofline': roofline_record(*prof, fused=fused), 'stages_s_per_step': stages,
    }
    if not terrain_headline and not args.no_other_mode:
        # the same workload through the other solve: assembled CSR + streaming SpMV (solve_non_fused, the path training needs)
        odt, _, ocfg, oprof, ostages, _ = run_cloud(2, 1, not fused)
        out['other_solve_mode'] = {'fused_mode': not fused, 'value': npts * 2 / odt, 'unit': 'points/s', 'ms_per_step': odt / 2 * 1e3, 'steps': 2,
                                   'warmup': 1, 'unknowns_M': ocfg['unknowns_M'], 'nnz_A': ocfg['nnz_A'], 'pcg_iters': ocfg['pcg_iters'],
                                   'roofline': roofline_record(*oprof, fused=not fused), 'stages_s_per_step': ostages}
        # north_star's KPI, in every line: the CSR SpMV roofline (target 0.70 of 8 TB/s), whichever solve the headline ran
        out['spmv_csr_roofline'] = out['other_solve_mode']['roofline'] if fused else out['roofline']
    if not terrain_headline and not args.no_scale_scene:
        # the same scene the N > 1 runs solve, on this one GPU (1 warm-up + 2 timed steps): the N=1 point of the curve
        torch.cuda.empty_cache()
        sdt, sn, scfg, sprof, sstages = run_terrain(2, 1, True)
        out['scale_scene'] = {'value': sn * 2 / sdt, 'unit': 'points/s', 'ms_per_step': sdt / 2 * 1e3, 'steps': 2, 'warmup': 1,
                              'config': scfg, 'roofline': roofline_record(*sprof, fused=True), 'stages_s_per_step': sstages}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and extra is not None:
        from oracle import waymo_cpu
        rec, xyz_np, nrm_np, scale = extra
        # bounded crop of the configs[2] cloud (model units), one per CPU worker
        c = xyz_np[0]
        idx = np.argsort(np.abs(xyz_np - c).max(1))[:args.cpu_sample]
        crop = os.path.join(tempfile.gettempdir(), 'nksr_bench_crop_%d.npz' % os.getpid())
        np.savez(crop, xyz=(xyz_np[idx] * np.float32(scale)).astype(np.float32), normal=nrm_np[idx], mise_iter=args.mise_iter)
        try:
            cb = waymo_cpu.measure(cores=args.cpu_cores or min(os.cpu_count() or 1, 64), repeats=args.cpu_repeats, crop=crop)
        finally:
            if os.path.exists(crop):
                os.remove(crop)
        # the GPU on the same input (same call sequence, sensor-only, through the product's preprocess_fn)
        d = np.load(waymo_cpu.BUNNY)
        bx = torch.from_numpy(d['xyz']).to(dev)
        bs = torch.from_numpy(waymo_cpu.synth_sensors(d['xyz'], d['normal'])).to(dev)
        fn = nksr_amd.get_estimate_normal_preprocess_fn(64, 85.0)

        def gpu_seq():
            f = rec.reconstruct(bx, sensor=bs, detail_level=None, approx_kernel_grad=True, solver_tol=1e-4, fused_mode=True, preprocess_fn=fn)
            return f.extract_dual_mesh(mise_iter=1)
        gpu_seq()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m = gpu_seq()
        torch.cuda.synchronize()
        cb['gpu_same_input'] = {'value': 5 * bx.shape[0] / (time.perf_counter() - t0), 'unit': 'points/s', 'mesh_triangles': int(m.f.shape[0]),
                                'note': 'one 10 000-point scan at a time on one MI355X: launch-latency bound, not a throughput figure'}
        out['cpu_baseline'] = cb
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
