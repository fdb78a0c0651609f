"""CPU: the marching-cubes restatement (oracle/mesh_oracle.py), the table the HIP library derives (host code, no
device needed) and the PLY writer."""
import numpy as np
import torch

from neural_graph_mapping_amd import _capi as K
from oracle import mesh_oracle as MO


def test_derived_table_is_complete_and_matches_the_library():
    count, table = MO.tables()
    assert count.max() == 5 and count[0] == 0 and count[255] == 0 and (count[1:255] > 0).all()
    tab = np.zeros(256 * 15, dtype=np.int8)
    cnt = np.zeros(256, dtype=np.int32)
    assert K.lib().ngm_marching_cubes_tables(tab.ctypes.data, cnt.ctypes.data) == 0
    assert (cnt == count).all() and (tab.reshape(256, 15) == table).all()


def test_oracle_marching_cubes_invariants():
    n = 24
    g = np.linspace(-1, 1, n, dtype=np.float32)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    v, f = MO.marching_cubes((0.6 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32), 0.0)
    h = 2 / (n - 1)
    closed, oriented, chi, area, vol = MO.mesh_stats(v * h - 1, f)
    assert closed and oriented and chi == 2
    assert abs(area - 4 * np.pi * 0.36) < 0.05 and abs(vol - 4 / 3 * np.pi * 0.216) < 0.02
    for seed in range(3):                                   # noise volumes hit every corner case incl. the ambiguous ones
        rng = np.random.default_rng(seed)
        vol_ = rng.standard_normal((12, 11, 10)).astype(np.float32)
        vol_[0] = vol_[-1] = -1; vol_[:, 0] = vol_[:, -1] = -1; vol_[:, :, 0] = vol_[:, :, -1] = -1
        v, f = MO.marching_cubes(vol_, 0.0)
        closed, oriented, _, _, _ = MO.mesh_stats(v, f)
        assert closed and oriented


def test_ply_round_trip(tmp_path):
    from neural_graph_mapping_amd import mesh as Mh
    v = torch.rand(7, 3)
    f = torch.randint(0, 7, (5, 3))
    c = torch.rand(7, 3) * 255
    Mh.save_ply(tmp_path / "a.ply", v, f, c)
    head = open(tmp_path / "a.ply", "rb").read(300).decode("ascii", errors="replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 7\nproperty float x")
    assert "property float red" in head and "property list uchar int vertex_index" in head
    v2, f2, c2 = Mh.load_ply(tmp_path / "a.ply")
    assert torch.equal(v, v2) and torch.equal(f, f2) and torch.equal(c, c2)
    Mh.save_ply(tmp_path / "b.ply", v, f)
    v3, f3, c3 = Mh.load_ply(tmp_path / "b.ply")
    assert torch.equal(v, v3) and torch.equal(f, f3) and c3 is None


def test_g17_extract_mesh_host_logic_vs_reference():
    """G17 = NeuralGraphMap._extract_mesh (rm.py:2186-2384) run for real with marching cubes stubbed by
    oracle/mesh_oracle.py: grid axes, block boundaries, the sign of the volume handed to marching cubes, and the map of
    its vertices back to the world, restated through the oracle's kNN evaluation and the host helpers of mesh.py."""
    import torch
    from conftest import load_golden, split_prefix
    from neural_graph_mapping_amd import mesh as Mh
    from oracle import ngm_oracle as O
    g = load_golden("g17_extract_mesh")
    r, res = float(g["field_radius"]), float(g["resolution"])
    axes = Mh.grid_axes(g["pos"], r, res)
    starts = [range(0, len(a) - 1, 200) for a in axes]                      # rm.py:2236-2240
    import itertools
    blocks = list(itertools.product(*starts))
    assert len(blocks) == int(g["num_blocks"])
    fs = O.FieldSpec(encoding="fourier", dim_enc=64, num_layers=2)
    params = {k: v for k, v in split_prefix(g, "p::").items() if k != "_neus_sd"}
    for b, (xs, ys, zs) in enumerate(blocks):
        bx, by, bz = axes[0][xs:xs + 201], axes[1][ys:ys + 201], axes[2][zs:zs + 201]
        assert (len(bx), len(by), len(bz)) == tuple(int(v) for v in g[f"b{b}_shape"])
        idx = g[f"b{b}_vol_idx"]
        iz = idx % len(bz); iy = (idx // len(bz)) % len(by); ix = idx // (len(bz) * len(by))
        pts = torch.stack((bx[ix], by[iy], bz[iz]), -1)
        out = O.field_set_forward_knn(pts, g["pos"], g["quat"], params, fs, radius=r, num_knn=2, distance_factor=10.0,
                                      outside_value=1.0)
        vol = -out[:, 3]                                                     # nrgbd: low_is_inside (rm.py:2283-2289)
        assert float((vol - g[f"b{b}_vol_val"]).abs().max()) < 3e-4
        if int(g[f"b{b}_num_verts"]):
            vw = Mh.grid_to_world(g[f"b{b}_vert_grid"], bx, by, bz)
            assert float((vw - g[f"b{b}_vert_world"]).abs().max()) < 1e-5
            # colours: second evaluation with radius + 0.1 as the INSIDE TEST only (rm.py:2324-2340, models.py:368-378)
            co = O.field_set_forward_knn(g[f"b{b}_vert_world"], g["pos"], g["quat"], params, fs, radius=r, num_knn=2,
                                         distance_factor=10.0, outside_value=1.0, mask_radius=r + 0.1)
            col = torch.clamp(co[:, :3], 0, 1) * 255
            assert float((col - g[f"b{b}_vert_color"]).abs().max()) <= 1.0
    assert float(g["isolevel"]) == 0.0
    assert float((torch.as_tensor(g["fields_txt"], dtype=torch.float32) - g["pos"]).abs().max()) < 1e-6   # identity transform
    assert int(g["ply_ascii"]) == 0 and int(g["ply_colors_as_uint8"]) == 0
