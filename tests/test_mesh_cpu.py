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
    assert "property float red" in head and "property list uchar int vertex_indices" in head
    v2, f2, c2 = Mh.load_ply(tmp_path / "a.ply")
    assert torch.equal(v, v2) and torch.equal(f, f2) and torch.equal(c, c2)
    Mh.save_ply(tmp_path / "b.ply", v, f)
    v3, f3, c3 = Mh.load_ply(tmp_path / "b.ply")
    assert torch.equal(v, v3) and torch.equal(f, f3) and c3 is None
