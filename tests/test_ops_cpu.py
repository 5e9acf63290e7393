"""Host-side logic of nksr_amd/ops.py that needs no GPU: the bit range every Morton-key sort of a reconstruction takes from the
cloud's bounding box (KeyBits / key_hint)."""
import itertools
import threading

import numpy as np


def _bits_needed(cells, bias):
    """Highest differing bit over a set of per-axis integer coordinates (interleaved x 3)."""
    b = 0
    for a in range(3):
        v = cells[:, a].astype(np.int64) + bias
        assert (v >= 0).all()
        b = max(b, int(v.min() ^ v.max()).bit_length())
        # (all integers between two bounds share their common binary prefix: the extremes decide)
    return 3 * b


def test_key_bits_cover_the_voxel_centres_of_coarse_levels_keyed_at_level_0():
    """KernelField._sorted_sites sorts the LEVEL-0 keys of normal sites that are voxel centres of levels < adaptive_depth; a
    27-neighbourhood voxel of level d reaches 1.5 * 2^d level-0 cells past the cloud's box.  The hinted bit range must cover them,
    also when the box ends a few cells below a power-of-two boundary that the coarse centres cross (round-4 advisor finding: a
    range too small leaves the radix sort partially sorted, silently)."""
    from nksr_amd.ops import KeyBits
    for depth, top, span in itertools.product((3, 4, 5, 6), (-4, -1, 60, 1020, 4092), (37, 700)):
        hi = [top, top - 3, top - 7]
        lo = [h - span for h in hi]
        kb = KeyBits(lo, hi, depth=depth)
        for d in range(depth):
            # the level-d cells of the box, their 27-neighbourhood, the centres of those voxels as level-0 cells
            cl = np.array([[(l >> d) - 1 for l in lo], [(h >> d) + 1 for h in hi]])
            centres0 = np.stack([(cl[0] << d) + ((1 << d) >> 1), (cl[1] << d) + ((1 << d) >> 1)])
            assert _bits_needed(centres0, 1 << 20) <= kb.bits(0), (depth, top, span, d)
            # and the level's own voxel keys at their own level
            assert _bits_needed(cl, 1 << (20 - d)) <= kb.bits(d), (depth, top, span, d)


def test_key_hint_is_per_thread():
    from nksr_amd import ops
    seen = {}

    def worker(name, kb):
        with ops.key_hint(kb):
            import time
            time.sleep(0.05)
            seen[name] = ops.varying_bits(None, level=0)
    a, b = ops.KeyBits([1000, 1000, 1000], [1010, 1010, 1010]), ops.KeyBits([1000, 1000, 1000], [100000, 1010, 1010])
    ts = [threading.Thread(target=worker, args=('a', a)), threading.Thread(target=worker, args=('b', b))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen['a'] == a.bits(0) and seen['b'] == b.bits(0) and seen['a'] != seen['b']
    assert getattr(ops._tls, 'hint', None) is None
