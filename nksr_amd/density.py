"""detail_level -> global scale (NKSR-USAGE.md:129-137: detail_level in [0,1], 0 = least detail /
most robust, 1 = most detail; voxel_size overrides it).  The reference derives the scale from
the point density inside the absent wheel; here [ASSUMPTION, DESIGN.md section 2.7] the finest
voxel is sized so that an occupied voxel holds ``ppv(detail) = 32 * (4/32)**detail`` points on
average (4 points per voxel at detail_level=1.0)."""
import torch

from . import ops
from ._lib import call, ptr, stream
from .svh import inv_w0_f32


def occupied_voxels(xyz, voxel_size):
    n = xyz.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device)
    call('nksr_point_keys', ptr(xyz), n, inv_w0_f32(voxel_size), ptr(keys), stream())
    return int(ops.sort_unique(keys).numel())


def scale_for_detail_level(xyz, detail_level, model_voxel_size, refine_iters=3):
    """One radix sort at a fine probe resolution gives the occupied-voxel count at every
    power-of-two multiple of it (Morton keys: coarser cell = key >> 3k); the target size is
    bracketed, log-interpolated and sharpened by a few bisection steps."""
    detail_level = min(max(detail_level, 0.0), 1.0)
    target = 32.0 * (4.0 / 32.0) ** detail_level
    n = xyz.shape[0]
    ext = float((xyz.max(0).values - xyz.min(0).values).max())
    if n < 8 or ext <= 0:
        return 1.0
    center = xyz.mean(0, keepdim=True)
    xc = (xyz - center).contiguous()     # keep |x / vs| small while probing tiny voxels
    vs0 = ext / 4096.0
    keys = torch.empty(n, dtype=torch.int64, device=xyz.device)
    call('nksr_point_keys', ptr(xc), n, inv_w0_f32(vs0), ptr(keys), stream())
    ks = ops.sort_keys(keys)
    counts = []
    for k in range(12):
        sh = ks >> (3 * k)
        counts.append(1 + int((sh[1:] != sh[:-1]).sum().item()))
    ppv = [n / c for c in counts]        # monotone non-decreasing in k
    lo, hi = vs0, vs0 * 2 ** 11
    for k in range(11):
        if ppv[k] < target <= ppv[k + 1]:
            lo, hi = vs0 * 2 ** k, vs0 * 2 ** (k + 1)
            break
    else:
        if target <= ppv[0]:
            return float(model_voxel_size) / vs0
    for _ in range(refine_iters):
        mid = (lo * hi) ** 0.5
        if n / max(occupied_voxels(xc, mid), 1) < target:
            lo = mid
        else:
            hi = mid
    vs = (lo * hi) ** 0.5
    return float(model_voxel_size) / vs
