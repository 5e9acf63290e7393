"""Integer conventions and the 1-D basis shared by every oracle module.

Every function here restates a piece of DESIGN.md section 2 ("the spec"), which is
itself the reconstruction of the absent ``nksr`` wheel described in SURVEY.md
Appendix B1/B2.  Reference anchors (call sites only; implementation absent):
  * level-d voxel size = voxel_size * 2**d, ``grids[d].active_grid_coords()`` are
    integer ijk (models/loss.py:36,45-46, models/nksr_net.py:57-62)
  * ``build_point_splatting`` activates the trilinear footprint of every point
    (models/nksr_net.py:62)
All integer decisions derive from ONE fp32 product  p = x * inv_w0  so that the
oracle and the HIP kernels agree bit-for-bit:
    H0 = floor(2 p)               (half-cell index at level 0)
    H_d = H0 >> d                 I_d = H_d >> 1      (cell containing x)
    S_d = (H_d - 1) >> 1          (base corner of the 8 nearest voxel centres)
"""
import numpy as np

COORD_BITS = 21
BIAS0 = 1 << 20  # level-0 bias; level d uses BIAS0 >> d so that key_d == key_0 >> 3d


def inv_w0_f32(voxel_size):
    """fp32 reciprocal of the finest voxel size (rounded once from double)."""
    return np.float32(1.0 / float(voxel_size))


def half_index(xyz, voxel_size):
    """H0 = floor(2 * (x * inv_w0)) as int32, product evaluated in fp32."""
    p = xyz.astype(np.float32) * inv_w0_f32(voxel_size)
    return np.floor(p * np.float32(2.0)).astype(np.int32), p


def _part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def _compact1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1249249249249249)
    v = (v | (v >> np.uint64(2))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v >> np.uint64(4))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v >> np.uint64(8))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v >> np.uint64(16))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v >> np.uint64(32))) & np.uint64(0x1FFFFF)
    return v


def morton_key(ijk, level):
    """63-bit Morton key of integer cell coordinates at ``level`` (x = lowest bit)."""
    b = ijk.astype(np.int64) + (BIAS0 >> level)
    assert (b >= 0).all() and (b < (1 << (COORD_BITS - level))).all(), "coordinate out of range"
    k = _part1by2(b[..., 0]) | (_part1by2(b[..., 1]) << np.uint64(1)) | (_part1by2(b[..., 2]) << np.uint64(2))
    return k.astype(np.int64)


def morton_decode(key, level):
    k = key.astype(np.uint64)
    x = _compact1by2(k)
    y = _compact1by2(k >> np.uint64(1))
    z = _compact1by2(k >> np.uint64(2))
    return (np.stack([x, y, z], -1).astype(np.int64) - (BIAS0 >> level)).astype(np.int32)


# 27-neighbour slot order: s = (dx+1)*9 + (dy+1)*3 + (dz+1)
NBR_OFFSETS = np.array([[dx, dy, dz] for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], np.int32)
# 8-corner order: c = cx*4 + cy*2 + cz
CORNER_OFFSETS = np.array([[cx, cy, cz] for cx in (0, 1) for cy in (0, 1) for cz in (0, 1)], np.int32)


def bspline3(u):
    """Quadratic B-spline weights of the 3 supporting centres (offset -1,0,+1) for
    local cell coordinate u in [0,1) and their derivatives d/du.  SURVEY.md App. B2."""
    u = u.astype(np.float32)
    one = np.float32(1.0)
    half = np.float32(0.5)
    w = np.stack([half * (one - u) * (one - u), np.float32(0.75) - (u - half) * (u - half), half * u * u], -1)
    dw = np.stack([u - one, np.float32(-2.0) * (u - half), u], -1)
    return w, dw
