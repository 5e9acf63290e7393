"""Thin tensor-level wrappers over the C-ABI primitives (sort / unique / scan / compaction /
hash).  PyTorch supplies device memory and the stream only."""
import threading

import torch

from ._lib import call, ptr, stream, with_tmp


def _bits(n):
    b = 1
    while (1 << b) < max(int(n), 1):
        b += 1
    return b


class KeyBits:
    """Bit range of the Morton keys of one cloud, from its bounding box -- known on the host after the ONE readback a reconstruction
    starts with, so no sort has to look at its keys first (round 3: ``int((keys ^ keys[:1]).max())`` before every large sort, a
    reduction + a host sync each).  lo / hi: level-0 cell coordinates of the box (inclusive).  The keys of level d are Morton codes of
    (cell >> d) + 2^(20 - d); everything a hierarchy level, a site list or a footprint stream of this cloud
    holds lies within two cells of the box at its level, and all integers between two bounds share their common binary prefix."""

    def __init__(self, lo, hi, depth=1):
        self.lo, self.hi = [int(v) for v in lo], [int(v) for v in hi]
        # Sites sorted under the hint include voxel centres of COARSE levels keyed at level 0 (KernelField._sorted_sites: the normal
        # sites of levels < adaptive_depth): a 27-neighbourhood voxel of level d lies up to 1.5 * 2^d level-0 cells past the box, so
        # the box is widened by two cells of the coarsest level (in level-0 cells, shifted down with the level) on top of the
        # two cells at the key's own level.  A bound too wide costs at most one radix pass; one too narrow a silently
        # half-sorted list.
        self.reach = 2 << max(int(depth) - 1, 0)

    def bits(self, level=0):
        b = 0
        for a in range(3):
            bias = 1 << (20 - level)
            lo = ((self.lo[a] - self.reach) >> level) - 2 + bias
            hi = ((self.hi[a] + self.reach) >> level) + 2 + bias
            b = max(b, (max(lo, 0) ^ hi).bit_length())
        return max(1, min(63, 3 * b))


_tls = threading.local()      # the hint is per thread: concurrent reconstructions must not see each other's boxes


class key_hint:
    """``with ops.key_hint(KeyBits(lo, hi)):`` -- sorts of Morton keys that name their ``level`` take their bit range from the box."""

    def __init__(self, kb):
        self.kb = kb

    def __enter__(self):
        self.prev = getattr(_tls, 'hint', None)
        _tls.hint = self.kb
        return self.kb

    def __exit__(self, *exc):
        _tls.hint = self.prev


def varying_bits(keys, level=None):
    """Number of low bits that differ anywhere in ``keys``: Morton keys of one cloud share their high bits
    (always when the cloud lies in one octant of the biased lattice), and every 8 constant bits save one
    radix pass.  With a key_hint in force and a ``level`` given, the answer comes from the cloud's box (no device work);
    otherwise small inputs are not worth the host round trip and large ones are looked at."""
    hint = getattr(_tls, 'hint', None)
    if level is not None and hint is not None:
        return hint.bits(level)
    if keys.numel() < (1 << 17):
        return 63
    return max(1, int((keys ^ keys[:1]).max()).bit_length())


def sort_keys(keys, end_bit=None, level=None):
    """Ascending radix sort of non-negative int64 keys (``level``: they are Morton keys of that hierarchy level, see key_hint)."""
    n = keys.numel()
    out = torch.empty_like(keys)
    if n:
        if end_bit is None:
            end_bit = varying_bits(keys, level)
        with_tmp('nksr_sort_keys_u64', keys.device, ptr(keys), ptr(out), n, 0, int(end_bit), stream())
    return out


def sort_pairs(keys, vals32, end_bit=None, pad=0, level=None):
    """Sort (int64 key, 32-bit payload) pairs by key.  ``pad`` extra (zeroed) elements are kept
    behind the returned views so that 16-byte loads may run past the end."""
    n = keys.numel()
    ko = torch.empty_like(keys)
    vo = torch.zeros(n + pad, dtype=vals32.dtype, device=vals32.device)[:n] if pad else torch.empty_like(vals32)
    if n:
        if end_bit is None:
            end_bit = varying_bits(keys, level)
        with_tmp('nksr_sort_pairs_u64_u32', keys.device, ptr(keys), ptr(ko), ptr(vals32), ptr(vo), n, 0, int(end_bit), stream())
    return ko, vo


def unique_sorted(keys_sorted):
    """Unique of an already sorted key stream (syncs to read the count)."""
    n = keys_sorted.numel()
    if n == 0:
        return keys_sorted
    out = torch.empty_like(keys_sorted)
    cnt = torch.zeros(1, dtype=torch.int64, device=keys_sorted.device)
    with_tmp('nksr_unique_u64', keys_sorted.device, ptr(keys_sorted), ptr(out), ptr(cnt), n, stream())
    return out[:int(cnt.item())].clone()


def sort_unique(keys, maybe_sorted=False, level=None):
    """``maybe_sorted``: the stream is expected to be strictly ascending already (keys derived in order from a sorted
    parent list): one comparison pass + host sync instead of the radix sort and the unique pass when it is."""
    if maybe_sorted and keys.numel() > 1 and bool((keys[1:] > keys[:-1]).all()):
        return keys
    return unique_sorted(sort_keys(keys, level=level))


def dedup_keys(keys):
    """A shorter stream with the same key set: repeats within runs of 2048 consecutive keys dropped (mode 3)."""
    n = keys.numel()
    raw = torch.empty(n, dtype=torch.int64, device=keys.device)
    cnt = torch.empty(1, dtype=torch.int64, device=keys.device)
    call('nksr_footprint_keys_dedup', None, ptr(keys), n, 0.0, 0, 3, ptr(raw), ptr(cnt), stream())
    return raw[:int(cnt.item())]


def dedup_corner_keys(cells):
    """The lattice corner keys of sorted dual cells with the repeats of neighbouring cells dropped (LDS hash set per
    workgroup, nksr_footprint_keys_dedup mode 2): same key SET as nksr_cell_corner_keys, 3-4x shorter stream."""
    nc = cells.numel()
    raw = torch.empty(nc * 8, dtype=torch.int64, device=cells.device)
    cnt = torch.empty(1, dtype=torch.int64, device=cells.device)
    call('nksr_footprint_keys_dedup', None, ptr(cells), nc, 0.0, 0, 2, ptr(raw), ptr(cnt), stream())
    return raw[:int(cnt.item())]


def exclusive_sum_i32(x):
    out = torch.empty_like(x)
    if x.numel():
        with_tmp('nksr_exclusive_sum_i32', x.device, ptr(x), ptr(out), x.numel(), stream())
    return out


def compact(flags):
    """Ordered indices (int32) of the non-zero int32 flags (syncs to read the count)."""
    n = flags.numel()
    if n == 0:
        return torch.empty(0, dtype=torch.int32, device=flags.device)
    nb = (n + 255) // 256
    counts = torch.empty(nb + 1, dtype=torch.int32, device=flags.device)
    counts[nb] = 0
    call('nksr_compact_block_counts', ptr(flags), n, ptr(counts), stream())
    offs = exclusive_sum_i32(counts)
    total = int(offs[nb].item())
    sel = torch.empty(total, dtype=torch.int32, device=flags.device)
    if total:
        call('nksr_compact_scatter', ptr(flags), n, ptr(offs), ptr(sel), stream())
    return sel


def sorted_lookup(sorted_keys, q):
    out = torch.empty(q.numel(), dtype=torch.int32, device=q.device)
    if q.numel():
        call('nksr_sorted_lookup', ptr(sorted_keys), sorted_keys.numel(), ptr(q), q.numel(), ptr(out), stream())
    return out


class HashTable:
    """Open-addressing hash  Morton key -> canonical voxel index."""

    def __init__(self, keys_sorted_unique):
        n = keys_sorted_unique.numel()
        cap = 16
        while cap < 2 * n:
            cap *= 2
        dev = keys_sorted_unique.device
        self.cap = cap
        # keys and values in ONE buffer cleared by ONE fill (-1 = all bytes 0xFF in both)
        buf = torch.full((cap * 12,), 255, dtype=torch.uint8, device=dev)
        self.hkeys = buf[:cap * 8].view(torch.int64)
        self.hvals = buf[cap * 8:].view(torch.int32)
        if n:
            call('nksr_hash_build', ptr(keys_sorted_unique), n, ptr(self.hkeys), ptr(self.hvals), cap, stream())

    def query(self, q):
        out = torch.empty(q.numel(), dtype=torch.int32, device=q.device)
        if q.numel():
            call('nksr_hash_query', ptr(q), q.numel(), ptr(self.hkeys), ptr(self.hvals), self.cap, ptr(out), stream())
        return out
