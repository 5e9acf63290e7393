"""Oracle: ``detail_level`` -> global scale.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference anchors (call sites; the rule itself lives in the absent wheel): ``reconstruct(xyz, normal,
detail_level=1.0)`` examples/recons_simple.py:26, recons_colored_mesh.py:27; ``detail_level=0.1`` gis_app.py:38-42;
semantics ("0 = least detail, 1 = most; voxel_size overrides it") NKSR-USAGE.md:129-137.  [ASSUMPTION, DESIGN.md
section 2.7] the finest voxel is sized so that an occupied voxel holds ``32 * (4/32)**detail`` points on average.

Restates nksr_amd/density.py step for step in numpy:
  1. exact fp32 bounding box, ``xc = xyz - lo`` (fp32), probe voxel ``vs0 = max extent / 4096`` (host double);
  2. occupied cells at every power-of-two multiple of vs0: unique counts of ``key >> 3k`` of the level-0 Morton keys
     (integer work: must agree EXACTLY with the device's sorted-XOR histogram);
  3. bracket of the target points-per-voxel, log-log interpolation inside it, ``refine_iters`` regula-falsi probes
     (each probe = one exact occupied-cell count at that voxel size), host double arithmetic -- same expressions, so the
     returned scale must agree to the last bit.
"""
import math

import numpy as np

from . import spec

PROBE_CELLS = 4096.0
LEVELS = 12


def _point_keys(xc, voxel_size):
    H0, _ = spec.half_index(xc, voxel_size)
    return spec.morton_key(H0 >> 1, 0)


def occupied_voxels(xc, voxel_size):
    return int(np.unique(_point_keys(xc, voxel_size)).shape[0])


def occupancy_counts(xyz):
    """(xc, vs0, counts[LEVELS]): occupied cells of width vs0 * 2**k, k = 0..LEVELS-1."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    lo, hi = xyz.min(0), xyz.max(0)
    ext = float((hi - lo).max())                    # fp32 subtraction, then widened
    xc = (xyz - lo[None]).astype(np.float32)
    if ext <= 0:
        return xc, 0.0, [1] * LEVELS
    vs0 = ext / PROBE_CELLS
    keys = _point_keys(xc, vs0)
    counts = [int(np.unique(keys >> (3 * k)).shape[0]) for k in range(LEVELS)]
    return xc, vs0, counts


def scale_for_detail_level(xyz, detail_level, model_voxel_size, refine_iters=2, trace=None):
    detail_level = min(max(float(detail_level), 0.0), 1.0)
    target = 32.0 * (4.0 / 32.0) ** detail_level
    n = xyz.shape[0]
    if n < 8:
        return 1.0
    xc, vs0, counts = occupancy_counts(xyz)
    if vs0 <= 0:
        return 1.0
    if trace is not None:
        trace['counts'], trace['vs0'], trace['probes'] = counts, vs0, []
    ppv = [n / c for c in counts]
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
