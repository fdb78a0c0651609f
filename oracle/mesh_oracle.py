"""CPU restatement of the mesh-extraction stage: marching cubes on a dense grid of field values.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE ***  (same rules as ngm_oracle.py)

PARITY UNPINNED against the reference's extractor: `_extract_mesh` (run_mapping.py:2255-2384) calls
`pytorch3d.ops.marching_cubes` (pyproject.toml:20, pytorch3d @ 47d5dc88; not vendored, not installed here).
What is restated is the published algorithm (Lorensen & Cline 1987: one vertex per grid edge the iso-surface
crosses, placed by linear interpolation; per cell a triangulation chosen by the 8-bit corner case) with the
triangulation table DERIVED from the cube topology instead of typed in:

  * corner i of a cell sits at offset (i & 1, i >> 1 & 1, i >> 2 & 1) along (x, y, z); a corner is "inside" when its
    value is > isolevel (the reference negates nrgbd / neus volumes first: low_is_inside, rm.py:2277-2289);
  * on each of the 6 faces the crossed edges are joined pairwise; a face with 4 crossed edges (inside corners on a
    diagonal) is resolved by cutting off each INSIDE corner separately -- a rule that only looks at the face's own 4
    values, so the two cells sharing the face agree and the surface is watertight;
  * the face segments of a cell form closed loops (every crossed edge lies on exactly 2 faces); each loop is fanned
    into triangles from its lowest-numbered edge and oriented so that normals point from inside to outside.

Vertex and face ORDER are defined here too (the HIP kernels reproduce them bit for bit): vertices in order of
(grid point linear index ((x * ny) + y) * nz + z, axis x < y < z) of the edge's lower end point; faces in order of
the cell's linear index, then the table's triangle order.
"""
import functools

import numpy as np

CORNER = np.array([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)])
# edge e joins corners EDGE[e]; numbering: 4 edges along x (axis 0), then 4 along y, then 4 along z, each ordered by
# the lower corner's index
EDGE = [(a, a | (1 << ax)) for ax in range(3) for a in range(8) if not a & (1 << ax)]
# 6 faces as corner cycles (walk around the square)
FACES = []
for ax in range(3):
    o1, o2 = [1 << b for b in range(3) if b != ax]
    for side in (0, 1):
        base = side << ax
        FACES.append([base, base | o1, base | o1 | o2, base | o2])
EDGE_ID = {frozenset(e): i for i, e in enumerate(EDGE)}


EDGE_FACES = [{fi for fi, f in enumerate(FACES) if set(e) <= set(f)} for e in EDGE]


def _all_triangulations(poly):
    """every triangulation of the (ordered) polygon, as lists of index triples, in a fixed enumeration order"""
    if len(poly) < 3:
        return [[]]
    if len(poly) == 3:
        return [[tuple(poly)]]
    res = []
    for k in range(1, len(poly) - 1):                    # the triangle on the polygon edge (poly[0], poly[-1])
        for left in _all_triangulations(poly[:k + 1]):
            for right in _all_triangulations(poly[k:]):
                res.append(left + [(poly[0], poly[k], poly[-1])] + right)
    return res


def _triangulate(loop):
    """A triangulation of the loop none of whose interior diagonals lies in a cube face: a diagonal between two edge
    points of the same face would put triangles INTO that face, where the neighbouring cell may do the same (a loop can
    visit an ambiguous face twice) -- overlapping coplanar triangles, a non-manifold edge.  First admissible one in the
    enumeration order; rotations of the loop keep the orientation."""
    n = len(loop)
    adjacent = {frozenset((loop[k], loop[(k + 1) % n])) for k in range(n)}
    for tri in _all_triangulations(loop):
        ok = True
        for t in tri:
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                if frozenset((a, b)) not in adjacent and EDGE_FACES[a] & EDGE_FACES[b]:
                    ok = False
        if ok:
            return tri
    raise AssertionError(f"no face-diagonal-free triangulation for loop {loop}")


@functools.lru_cache(maxsize=None)
def tables():
    """(tri_count[256], tri_table[256, MAX_TRI * 3] of edge ids, -1 padded)."""
    out = []
    for case in range(256):
        inside = [(case >> i) & 1 for i in range(8)]
        nbr = {}                                                   # crossed edge -> its (two) neighbours along face segments
        for f in FACES:
            cross = [EDGE_ID[frozenset((f[k], f[(k + 1) % 4]))] for k in range(4)
                     if inside[f[k]] != inside[f[(k + 1) % 4]]]
            if len(cross) == 2:
                pairs = [tuple(cross)]
            elif len(cross) == 4:
                # two inside corners on a diagonal: each inside corner is cut off by a segment between ITS two edges
                pairs = []
                for k in range(4):
                    if inside[f[k]]:
                        pairs.append((EDGE_ID[frozenset((f[k], f[(k + 1) % 4]))], EDGE_ID[frozenset((f[k], f[(k - 1) % 4]))]))
            else:
                pairs = []
            for a, b in pairs:
                nbr.setdefault(a, []).append(b)
                nbr.setdefault(b, []).append(a)
        tris, seen = [], set()
        for start in sorted(nbr):
            if start in seen:
                continue
            loop, prev, cur = [start], None, start                  # every crossed edge has exactly two neighbours
            seen.add(start)
            while True:
                a, b = nbr[cur]
                nxt = b if a == prev else a
                if nxt == start:
                    break
                loop.append(nxt)
                seen.add(nxt)
                prev, cur = cur, nxt
            # orientation: Newell normal of the loop (edge mid points) against the mean inside -> outside direction
            mid = np.array([(CORNER[EDGE[e][0]] + CORNER[EDGE[e][1]]) / 2.0 for e in loop])
            nrm = np.zeros(3)
            for k in range(len(loop)):
                p, q = mid[k], mid[(k + 1) % len(loop)]
                nrm += np.cross(p, q)
            d = np.zeros(3)
            for e in loop:
                a, b = EDGE[e]
                d += (CORNER[b] - CORNER[a]) * (1 if inside[a] else -1)
            if np.dot(nrm, d) < 0:
                loop = [loop[0]] + loop[:0:-1]
            tris += _triangulate(loop)
        out.append(tris)
    max_tri = max(len(t) for t in out)
    table = -np.ones((256, max_tri * 3), dtype=np.int8)
    count = np.zeros(256, dtype=np.int32)
    for c, t in enumerate(out):
        count[c] = len(t)
        table[c, :3 * len(t)] = np.array(t, dtype=np.int8).reshape(-1) if t else []
    return count, table


def marching_cubes(volume: np.ndarray, isolevel: float):
    """volume (nx, ny, nz) float32 -> (verts (V, 3) float32 in grid-index coordinates, faces (T, 3) int64)."""
    vol = np.asarray(volume, dtype=np.float32)
    nx, ny, nz = vol.shape
    iso = np.float32(isolevel)
    inside = vol > iso
    count, table = tables()
    vid = -np.ones((nx, ny, nz, 3), dtype=np.int64)
    flags = np.zeros((nx, ny, nz, 3), dtype=bool)
    flags[:-1, :, :, 0] = inside[:-1] != inside[1:]
    flags[:, :-1, :, 1] = inside[:, :-1] != inside[:, 1:]
    flags[:, :, :-1, 2] = inside[:, :, :-1] != inside[:, :, 1:]
    vid[flags] = np.arange(int(flags.sum()))
    idx = np.argwhere(flags)                                          # row-major: (x, y, z, axis) ascending = vertex order
    p0 = idx[:, :3]
    p1 = p0.copy()
    p1[np.arange(len(idx)), idx[:, 3]] += 1
    v0, v1 = vol[tuple(p0.T)], vol[tuple(p1.T)]
    t = ((iso - v0) / (v1 - v0)).astype(np.float32)                   # fp32, this op order (the kernels use the same)
    verts = p0.astype(np.float32)
    verts[np.arange(len(idx)), idx[:, 3]] += t
    case = np.zeros((nx - 1, ny - 1, nz - 1), dtype=np.int32)
    for i in range(8):
        ox, oy, oz = CORNER[i]
        case |= inside[ox:nx - 1 + ox, oy:ny - 1 + oy, oz:nz - 1 + oz].astype(np.int32) << i
    faces = []
    cells = np.argwhere(count[case] > 0)
    for x, y, z in cells:
        c = case[x, y, z]
        for k in range(count[c]):
            tri = []
            for e in table[c, 3 * k:3 * k + 3]:
                a, _ = EDGE[e]
                ax = e // 4
                tri.append(vid[x + CORNER[a][0], y + CORNER[a][1], z + CORNER[a][2], ax])
            faces.append(tri)
    return verts, np.array(faces, dtype=np.int64).reshape(-1, 3)


def mesh_stats(verts, faces):
    """(every undirected edge used exactly twice, consistently oriented, Euler characteristic, area, signed volume)."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    und = np.sort(e, 1)
    _, cnt = np.unique(und, axis=0, return_counts=True)
    closed = bool((cnt == 2).all())
    _, dcnt = np.unique(e, axis=0, return_counts=True)                 # a directed edge may appear only once
    oriented = bool((dcnt == 1).all())
    chi = len(np.unique(faces)) - len(cnt) + len(faces)
    a, b, c = verts[faces[:, 0]].astype(np.float64), verts[faces[:, 1]].astype(np.float64), verts[faces[:, 2]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    vol = (a * np.cross(b, c)).sum() / 6.0
    return closed, oriented, int(chi), float(area), float(vol)
