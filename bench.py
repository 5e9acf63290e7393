#!/usr/bin/env python
"""Headline benchmark (driver contract):  python bench.py --gpus N --steps K --warmup W

metric  : reconstructed points/sec (solve + mesh), BASELINE.json
workload: BASELINE.json configs[2] -- synthetic 1M-point oriented cloud per GPU,
          detail_level=1.0, reconstruct() + extract_dual_mesh(mise_iter=1).  One "step" = one
          full pass of the hot path over the rank's resident cloud.  Weak scaling: every rank
          owns one 40x40x10 tile of the scene (tiles adjacent along x).
roofline: the CG SpMV (csrc/pcg.hip k_spmv), algorithmic bytes 8*nnz + 12*M + 4 per launch
          divided by the average launch duration measured live with HIP events on the solve
          stream inside the timed region (nksr_pcg_profile).
cpu_baseline: the CPU oracle ("port" -- the reference's own CPU path is the absent wheel) on a
          bounded spatial crop of the same workload, timed on rank 0 at N=1.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver (RCCL peer mappings)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3e12 achievable)


def load_traffic(bytes_per_launch=None):
    """HBM bytes per SpMV launch from the committed rocprofv3 --pmc pass; None when the committed
    pass was taken on a different matrix than this run's."""
    best = None
    pdir = os.path.join(ROOT, 'profiles')
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith('_spmv_pmc.json'):
                try:
                    rec = json.load(open(os.path.join(pdir, f)))
                    if bytes_per_launch is None or abs(rec.get('algorithmic_bytes_per_launch', 0) - bytes_per_launch) < 0.02 * bytes_per_launch:
                        best = rec.get('hbm_bytes_per_launch')
                except Exception:
                    pass
    return best


def cpu_baseline(xyz, nrm, scale, n_sample, mise_iter, net_params=None):
    """Oracle pipeline on a spatial crop holding ~n_sample points (same density as the GPU run)."""
    from oracle import pipeline
    c = xyz[0]
    d = np.abs(xyz - c).max(1)
    idx = np.argsort(d)[:n_sample]
    xs = (xyz[idx] * np.float32(scale)).astype(np.float32)
    ns = nrm[idx]
    t0 = time.perf_counter()
    timing = {}
    fld = pipeline.reconstruct(xs, ns, tol=1e-5, timing=timing, net_params=net_params)
    v, f = pipeline.extract_dual_mesh(fld, mise_iter=mise_iter)
    dt = time.perf_counter() - t0
    return {'value': len(idx) / dt, 'unit': 'points/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle.pipeline reconstruct+extract_dual_mesh(mise_iter=%d) on a %d-point spatial crop of the '
                      'same cloud at the same scale: %.1fs (M=%d nnz=%d iters=%d)' % (
                          mise_iter, len(idx), dt, timing.get('M', 0), timing.get('nnz', 0), timing.get('iters', 0))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--points', type=int, default=1_000_000, help='points per GPU')
    ap.add_argument('--mise-iter', type=int, default=1)
    ap.add_argument('--detail-level', type=float, default=1.0)
    ap.add_argument('--cpu-sample', type=int, default=20000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
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
    from nksr_amd import solver, utils

    extent = (40.0, 40.0, 10.0)
    rec = nksr_amd.Reconstructor(dev)
    rec.sync_timing = True
    stage_acc = {}
    if world == 1:
        xyz_np, nrm_np = utils.synth_scene(args.points, seed=0, extent=extent, noise=0.01)
        xyz = torch.from_numpy(xyz_np).to(dev)
        nrm = torch.from_numpy(nrm_np).to(dev)
        chunk_size = None
    else:
        # N tiles side by side along x, every rank holds the full cloud (reference semantics: one
        # reconstruct() call over the scene); chunk_size = tile width => one chunk per rank.  The
        # detail_level scale of the N=1 run is applied up front because chunk mode takes a pre-scaled
        # cloud (NKSR-USAGE.md:137).
        tiles = [utils.synth_scene(args.points, seed=r, extent=extent, noise=0.01, origin=(r * extent[0], 0.0, 0.0))
                 for r in range(world)]
        from nksr_amd.density import scale_for_detail_level
        scale = scale_for_detail_level(torch.from_numpy(tiles[0][0]).to(dev), args.detail_level, rec.hparams.voxel_size)
        xyz_np = np.concatenate([t[0] for t in tiles]) * np.float32(scale)
        xyz_np[:, 0] -= xyz_np[:, 0].min()
        nrm_np = np.concatenate([t[1] for t in tiles])
        xyz = torch.from_numpy(xyz_np.astype(np.float32)).to(dev)
        nrm = torch.from_numpy(nrm_np).to(dev)
        chunk_size = float(xyz_np[:, 0].max()) / world + 1e-3

    def step():
        if chunk_size is None:
            field = rec.reconstruct(xyz, nrm, detail_level=args.detail_level)
        else:
            field = rec.reconstruct(xyz, nrm, detail_level=None, chunk_size=chunk_size)
        t0 = time.perf_counter()
        mesh = field.extract_dual_mesh(mise_iter=args.mise_iter)
        torch.cuda.synchronize()
        tm = time.perf_counter() - t0
        for k, v in list(rec.timing.items()) + [('t_mesh', tm)]:
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        return field, mesh

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_acc.clear()
    solver.profile_spmv(True)
    fence()
    t0 = time.perf_counter()
    step_ends = []
    field = mesh = None
    for _ in range(args.steps):
        # the previous result is released first: every step then has the memory footprint of the warmup step
        # and the caching allocator serves it from its pool (holding the old field while building the new one
        # made the second timed step pay ~40 ms of fresh hipMalloc)
        field = mesh = None
        field, mesh = step()
        step_ends.append(time.perf_counter())     # step() ends with a device sync
    fence()
    dt = time.perf_counter() - t0
    if os.environ.get('NKSR_BENCH_VERBOSE'):
        print('per-step ms:', [round((b - a) * 1e3, 1) for a, b in zip([t0] + step_ends[:-1], step_ends)], file=sys.stderr)
    spmv_ms, spmv_launches = solver.profile_spmv(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    info = field.solve_info if world == 1 else [f for f in field.fields.values() if f.solve_info][0].solve_info
    M, nnz = info['M'], info['nnz']
    b_spmv = 8.0 * nnz + 12.0 * M + 4.0
    avg_s = (spmv_ms / max(spmv_launches, 1)) * 1e-3
    achieved = b_spmv / avg_s if avg_s > 0 else 0.0
    total_points = args.points * world * args.steps
    out = {
        'metric': 'reconstructed points/sec (solve+mesh)', 'value': total_points / dt, 'unit': 'points/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[2]: synthetic %d-point oriented cloud per GPU (8 spheres/tori in a 40x40x10 tile, '
                               'sigma=0.01), detail_level=%.1f, reconstruct()+extract_dual_mesh(mise_iter=%d)' % (
                                   args.points, args.detail_level, args.mise_iter),
                   'points_per_gpu': args.points, 'tree_depth': rec.hparams.tree_depth, 'kernel_dim': rec.hparams.kernel_dim,
                   'unknowns_M': M, 'nnz_A': nnz, 'pcg_iters': info['iters'], 'pcg_rel_residual': info['rel_residual'],
                   'mesh_vertices': int(mesh.v.shape[0]), 'mesh_triangles': int(mesh.f.shape[0]),
                   'global_scale': field.scale if world == 1 else scale,
                   'parallelism': 'none' if world == 1 else 'chunks sharded 1/rank, all_gather of solved fields before meshing, mesh gather'},
        'roofline': {'bound': 'hbm', 'kernel': 'k_spmv<0> + k_spmv_fixup (CSR SpMV inside the PCG loop)',
                     'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK,
                     'traffic': load_traffic(b_spmv), 'bytes_per_launch': b_spmv, 'avg_launch_us': avg_s * 1e6,
                     'launches_timed': spmv_launches},
        'stages_s_per_step': {k: v / args.steps for k, v in sorted(stage_acc.items())},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import network as onet
        out['cpu_baseline'] = cpu_baseline(xyz_np, nrm_np, field.scale, args.cpu_sample, args.mise_iter,
                                           onet.export_params(rec.network))
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
