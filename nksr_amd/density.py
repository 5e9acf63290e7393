"""detail_level -> global scale (NKSR-USAGE.md:129-137: detail_level in [0,1], 0 = least detail /
most robust, 1 = most detail; voxel_size overrides it).  The reference derives the scale from
the point density inside the absent wheel; here [ASSUMPTION, DESIGN.md section 2.7] the finest
voxel is sized so that an occupied voxel holds ``ppv(detail) = 32 * (4/32)**detail`` points on
average (4 points per voxel at detail_level=1.0)."""
import torch

from . import ops
from ._lib import call, lib, ptr, stream
from .svh import inv_w0_f32


def occupied_voxels(xyz, voxel_size):
    n = xyz.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device)
    call('nksr_point_keys', ptr(xyz), n, inv_w0_f32(voxel_size), ptr(keys), stream())
    return int(ops.sort_unique(keys).numel())


def bbox_center(xyz):
    """(lo[3], hi[3], None): exact bounds of the cloud (nksr_bbox: two small kernels; torch's reductions of an [N,3]
    tensor -- along either axis -- take 170 us each at 1 M points).  The third slot used to carry the mean; nothing reads it."""
    if xyz.device.type != 'cuda':
        xt = xyz.t().contiguous()
        return xt.amin(1), xt.amax(1), None
    xyz = xyz.contiguous()
    nw = int(lib.nksr_bbox_work_floats())
    work = torch.empty(nw + 6, dtype=torch.float32, device=xyz.device)
    call('nksr_bbox', ptr(xyz), xyz.shape[0], ptr(work), ptr(work[nw:]), stream())
    return work[nw:nw + 3], work[nw + 3:], None


PROBE_CELLS = 4096.0
LEVELS = 12


def occupancy_counts(xyz):
    """(xc, vs0, counts[LEVELS]): the cloud moved to its bounding-box corner, the probe voxel size (max extent / 4096) and the
    number of occupied cells of width vs0 * 2**k for k = 0..11 -- ONE radix sort of the probe-level Morton keys: a coarser cell
    is ``key >> 3k`` and adjacent sorted keys differ first at level floor(log8(a ^ b)), so a histogram of that level over the
    sorted stream gives every count (integer work, exact; restated in oracle/density.py)."""
    n = xyz.shape[0]
    lo3, hi3, _ = bbox_center(xyz)
    ext = float((hi3 - lo3).max())
    xc = (xyz - lo3[None]).contiguous()        # small non-negative coordinates: short Morton keys, fewer radix passes
    if ext <= 0:
        return xc, 0.0, [1] * LEVELS
    vs0 = ext / PROBE_CELLS
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device)
    call('nksr_point_keys', ptr(xc), n, inv_w0_f32(vs0), ptr(keys), stream())
    ks = ops.sort_keys(keys)
    thr = torch.tensor([8 ** k for k in range(LEVELS)], dtype=torch.int64, device=xyz.device)
    lvl = torch.bucketize(ks[1:] ^ ks[:-1], thr, right=True)       # 0 = equal keys, j = differ below level j
    hist = torch.bincount(lvl, minlength=LEVELS + 1).tolist()
    return xc, vs0, [1 + sum(hist[k + 1:]) for k in range(LEVELS)]


def scale_for_detail_level(xyz, detail_level, model_voxel_size, refine_iters=2, trace=None):
    """The target voxel size is bracketed between two power-of-two multiples of the probe size (occupancy_counts),
    log-interpolated inside the bracket and sharpened by ``refine_iters`` regula-falsi probes (one sort each).
    ``trace`` (dict): receives the level counts and the probes (parity tests)."""
    import math
    detail_level = min(max(detail_level, 0.0), 1.0)
    target = 32.0 * (4.0 / 32.0) ** detail_level
    n = xyz.shape[0]
    if n < 8:
        return 1.0
    xc, vs0, counts = occupancy_counts(xyz)
    if vs0 <= 0:
        return 1.0
    if trace is not None:
        trace['counts'], trace['vs0'], trace['probes'] = counts, vs0, []
    ppv = [n / c for c in counts]        # monotone non-decreasing in k
    for k in range(LEVELS - 1):
        if ppv[k] < target <= ppv[k + 1]:
            break
    else:
        if target <= ppv[0]:
            return float(model_voxel_size) / vs0
        k = LEVELS - 2
    lo, hi, plo, phi = vs0 * 2 ** k, vs0 * 2 ** (k + 1), ppv[k], ppv[k + 1]
    vs = (lo * hi) ** 0.5
    for it in range(refine_iters + 1):
        # points-per-voxel is close to a power law in the voxel size: interpolate in log-log
        if phi > plo and plo > 0:
            t = (math.log(target) - math.log(plo)) / (math.log(phi) - math.log(plo))
            vs = lo * (hi / lo) ** min(max(t, 0.02), 0.98)
        else:
            vs = (lo * hi) ** 0.5
        if it == refine_iters:
            break
        occ = occupied_voxels(xc, vs)
        if trace is not None:
            trace['probes'].append((vs, occ))
        p = n / max(occ, 1)
        if p < target:
            lo, plo = vs, p
        else:
            hi, phi = vs, p
    return float(model_voxel_size) / vs
