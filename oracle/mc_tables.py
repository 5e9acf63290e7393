"""Oracle: marching-cubes case table, generated (not transcribed).

The reference extracts meshes with dual marching cubes (field.extract_dual_mesh,
examples/recons_simple.py:27, models/nksr_net.py:214,284); its 256-case table lives
in the absent wheel.  This module derives a table from first principles so that
the topology rule is explicit and checkable:
  * corner c = cx*4 + cy*2 + cz, "inside" iff f > 0 (f>0 inside: models/loss.py:99-100)
  * edge e = axis*4 + o1*2 + o2 (o1,o2 = coordinates of the two other axes, ascending)
  * on every cube face the crossing edges are paired so that each maximal run of
    inside corners (walking the face counter-clockwise seen from outside) is cut
    off by one segment; ambiguous faces therefore always isolate inside corners.
    The rule depends only on the 4 face-corner states, hence neighbouring cells
    agree on every shared face and the extracted surface is watertight.
  * segments are directed (entry edge -> exit edge of the run) which orients every
    loop so that triangle normals point from inside (f>0) to outside.
  * each loop is fan-triangulated from the smallest edge id whose fan has no diagonal joining
    two edges of one cube face (such a diagonal would lie IN that face and be shared with the
    neighbouring cell's triangles: a non-manifold edge); such a start always exists.
"""
import numpy as np

CYCLIC = {0: (1, 2), 1: (2, 0), 2: (0, 1)}


def corner_index(c):
    return c[0] * 4 + c[1] * 2 + c[2]


def edge_id(axis, coords):
    others = [a for a in range(3) if a != axis]
    return axis * 4 + coords[others[0]] * 2 + coords[others[1]]


def edge_corners(e):
    axis, r = divmod(e, 4)
    others = [a for a in range(3) if a != axis]
    lo = [0, 0, 0]
    lo[others[0]] = r >> 1
    lo[others[1]] = r & 1
    hi = list(lo)
    hi[axis] = 1
    return corner_index(lo), corner_index(hi), axis, lo


def _edge_between(c0, c1):
    axis = [a for a in range(3) if c0[a] != c1[a]]
    assert len(axis) == 1
    return edge_id(axis[0], c0)


def face_cycles():
    faces = []
    for a in range(3):
        b, c = CYCLIC[a]
        for s in (0, 1):
            ring = [(0, 0), (1, 0), (1, 1), (0, 1)]
            if s == 0:
                ring = [(0, 0), (0, 1), (1, 1), (1, 0)]
            corners = []
            for (pb, pc) in ring:
                v = [0, 0, 0]
                v[a] = s
                v[b] = pb
                v[c] = pc
                corners.append(tuple(v))
            faces.append(corners)
    return faces


def case_loops(config):
    """Directed edge loops of one sign configuration."""
    nxt = {}
    for corners in face_cycles():
        ins = [(config >> corner_index(v)) & 1 for v in corners]
        if sum(ins) in (0, 4):
            continue
        for i in range(4):
            # run of inside corners starting at corner i+1 (entered over edge i)
            if not ins[i] and ins[(i + 1) % 4]:
                e_in = _edge_between(corners[i], corners[(i + 1) % 4])
                j = (i + 1) % 4
                while ins[(j + 1) % 4]:
                    j = (j + 1) % 4
                e_out = _edge_between(corners[j], corners[(j + 1) % 4])
                assert e_in not in nxt
                nxt[e_in] = e_out
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start, "open loop"
        loops.append(loop)
    return loops


def edge_faces(e):
    axis, r = divmod(e, 4)
    others = [a for a in range(3) if a != axis]
    return {(others[0], r >> 1), (others[1], r & 1)}


def fan_start(loop):
    """Rotation (as index into loop) of the canonical fan start."""
    n = len(loop)
    for start in sorted(range(n), key=lambda i: loop[i]):
        l = loop[start:] + loop[:start]
        if not any(edge_faces(l[0]) & edge_faces(l[i]) for i in range(2, n - 1)):
            return start
    raise AssertionError('no manifold fan for loop %s' % (loop,))


def build_tables():
    tris = []
    for config in range(256):
        t = []
        for loop in case_loops(config):
            k = fan_start(loop)
            loop = loop[k:] + loop[:k]
            for i in range(1, len(loop) - 1):
                t.append((loop[0], loop[i], loop[i + 1]))
        tris.append(t)
    max_t = max(len(t) for t in tris)
    count = np.array([len(t) for t in tris], np.int32)
    table = np.full((256, max_t, 3), -1, np.int32)
    for c, t in enumerate(tris):
        for i, tri in enumerate(t):
            table[c, i] = tri
    return count, table


TRI_COUNT, TRI_TABLE = build_tables()
EDGE_LO = np.array([edge_corners(e)[0] for e in range(12)], np.int32)
EDGE_HI = np.array([edge_corners(e)[1] for e in range(12)], np.int32)
EDGE_AXIS = np.array([edge_corners(e)[2] for e in range(12)], np.int32)
EDGE_LO_OFF = np.array([edge_corners(e)[3] for e in range(12)], np.int32)
