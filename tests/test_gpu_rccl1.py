"""RCCL itself, on the ONE GPU of the test box: a single-rank ``nccl`` (= RCCL on ROCm) process group with
NKSR_DIST_FORCE=1, which makes every collective helper of nksr_amd/dist.py run through the backend instead of
short-circuiting at world size 1 (the first time RCCL touches these buffers must not be the driver's
8-GPU run).  Exercised: all_reduce (bounding box, per-core counts), all_gather_into_tensor on int64 size vectors and on the
uint8 payload buffer of the chunk-halo exchange, the mesh gather -- and the distributed chunk pipeline end to end with
sharded input, which must reproduce the plain single-process mesh bit for bit.  Round 4: the neighbour-only halo exchange
(all_to_all_single with per-pair split sizes) and the mesh gather's point-to-point transport (batch_isend_irecv, as a send to
self) run here under RCCL too; with a second rank they are covered by the gloo tests (tests/test_dist_cpu.py, tests/test_gpu_dist2.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from nksr_amd import utils
    xyz, nrm = utils.synth_scene(120000, seed=7, extent=(24.0, 12.0, 6.0), noise=0.0, n_objects=6)
    return (xyz - xyz.min(0)).astype(np.float32), nrm


def _canon(mesh):
    """Index-free form (tests/test_gpu_dist2.py): the gather re-numbers the vertices (seam merge: sorted by canonical id)."""
    v, f = mesh.v.cpu().numpy(), mesh.f.cpu().numpy()
    vb = np.ascontiguousarray(v).view(np.dtype((np.void, 12))).ravel()
    tri = np.ascontiguousarray(v[f].reshape(len(f), 9))
    tb = tri.view(np.dtype((np.void, 36))).ravel()
    return v[np.argsort(vb, kind='stable')], tri[np.argsort(tb, kind='stable')]


def _same(a, b):
    (va, ta), (vb, tb) = _canon(a), _canon(b)
    return va.shape == vb.shape and ta.shape == tb.shape and np.array_equal(va, vb) and np.array_equal(ta, tb)


def _worker(port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    try:
        torch.cuda.set_device(0)
        import nksr_amd
        from nksr_amd import dist as D
        dev = torch.device('cuda:0')
        xyz, nrm = _scene()
        t = lambda a: torch.from_numpy(a).to(dev)
        rec = nksr_amd.Reconstructor(dev)
        plain = rec.reconstruct(t(xyz), t(nrm), detail_level=None, chunk_size=8.1)
        m0 = plain.extract_dual_mesh(mise_iter=1)
        assert not D.active() and len(plain.fields) >= 3
        os.environ['NKSR_DIST_FORCE'] = '1'
        dist.init_process_group('nccl', rank=0, world_size=1)
        assert D.active() and D.world() == (0, 1)
        # the helpers, on device buffers of every dtype they carry
        g = torch.Generator().manual_seed(0)
        parts = [torch.randint(0, 1 << 40, (1001,), generator=g, dtype=torch.int64).to(dev), torch.randn(777, generator=g).to(dev),
                 torch.randint(0, 255, (13,), generator=g, dtype=torch.uint8).to(dev), torch.zeros(0, dtype=torch.int8, device=dev)]
        got = D.all_gather_tensors(parts)
        assert len(got) == 1 and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(got[0], parts))
        pay = D.exchange_payloads({3: (parts[0], parts[1]), 9: (parts[0][:5].contiguous(), parts[1][:0].contiguous())}, [3, 9])
        assert torch.equal(pay[3][0], parts[0]) and torch.equal(pay[3][1], parts[1]) and pay[9][1].numel() == 0
        # the neighbour-only exchange (all_to_all_single with per-pair split sizes) and the mesh gather's point-to-point transport
        # (batch_isend_irecv: a send to self inside one group call under the forced single-rank group)
        pay2 = D.exchange_payloads_to({3: (parts[0], parts[1]), 9: (parts[0][:5].contiguous(), parts[1][:0].contiguous())}, {3: [0]})
        assert sorted(pay2) == [3, 9] and torch.equal(pay2[3][0], parts[0]) and torch.equal(pay2[3][1], parts[1]) and pay2[9][0].numel() == 5
        rv = D.all_to_all_tensors([[parts[0], parts[1], parts[2]]])
        assert len(rv) == 1 and all(torch.equal(a, b) for a, b in zip(rv[0], parts[:3]))
        gt = D.gather_tensors([parts[0], parts[1], parts[2], parts[3]])
        assert len(gt) == 1 and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(gt[0], parts))
        # the distributed chunk pipeline through RCCL: sharded input (bbox / count all_reduce), halo exchange, mesh gather
        lo, hi = xyz.min(0), xyz.max(0)
        fld = rec.reconstruct(t(xyz), t(nrm), detail_level=None, chunk_size=8.1, sharded_input=True)
        assert fld.distributed and sorted(fld.fields) == sorted(plain.fields)
        m1 = fld.extract_dual_mesh(mise_iter=1)
        assert _same(m0, m1), 'mesh through the RCCL path differs from the plain one'
        fld2 = rec.reconstruct(t(xyz), t(nrm), detail_level=None, chunk_size=8.1, sharded_input=True,
                               chunk_bounds=([float(v) for v in lo], [float(v) for v in hi]), chunk_owner=[0] * len(plain.cores))
        m2 = fld2.extract_dual_mesh(mise_iter=1)
        assert _same(m0, m2)
        # the adaptive dual graph's pieces through the same transport (dist.gather_named: five-word vertex names)
        plain.dual_graph = fld2.dual_graph = 'adaptive'
        a0, a2 = plain.extract_dual_mesh(mise_iter=1), fld2.extract_dual_mesh(mise_iter=1)
        assert a0.f.shape[0] > 1000 and _same(a0, a2) and a2.vertex_names5.shape == (a2.v.shape[0], 5)
        q.put('ok')
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put(repr(e) + traceback.format_exc()[-2500:])
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def test_single_rank_rccl_runs_the_whole_distributed_path():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert res == 'ok', res
