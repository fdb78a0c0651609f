"""Stress of the kNN evaluation path against dependence on stale workspace contents: before every call the caching allocator's
free blocks are filled with random bits (the workspace is carved from them), map sizes and layouts vary, every result is compared
with a brute-force torch evaluation of the neighbour search (indices through the blend weights of constant-output fields)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_graph_mapping_amd import _capi as K  # noqa: E402
from neural_graph_mapping_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def poison(nbytes=1 << 30):
    t = torch.randint(-2 ** 31, 2 ** 31 - 1, (nbytes // 4,), device=DEV, dtype=torch.int32)
    del t


def main(iterations=40):
    torch.manual_seed(0)
    fc = K.field_cfg(encoding="fourier", dim_enc=32, num_layers=1)
    bad = 0
    for it in range(iterations):
        NF = [50, 700, 10000, 40000, 3][it % 5]
        Kn = [2, 1, 4, 2, 3][it % 5]
        P = [6000, 50000, 4000, 3000, 999][it % 5]
        n = int(round(NF ** (1 / 3))) + 1
        g = torch.arange(n) * (2 / 3 ** 0.5)
        pos = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)[torch.randperm(n ** 3)[:NF]]
        pos = (pos + 1e-3 * torch.randn(pos.shape[0], 3)).to(DEV)
        NF = pos.shape[0]
        quat = torch.zeros(NF, 4, device=DEV); quat[:, 0] = 1
        pts = (torch.rand(P, 3) * (float(g[-1]) + 4) - 2).to(DEV)
        # fields with constant outputs: zero weights, output bias = (field index + 1) * 1e-3 on every channel
        shp = K.param_shapes(fc)
        params = {k: torch.zeros(NF, *s, device=DEV) for k, s in shp.items()}
        last_b = [k for k in shp if k.endswith(".bias")][-1]
        val = ((torch.arange(NF, device=DEV) % 997) + 1).float() * 1e-3
        params[last_b] += val[:, None]
        poison()
        out = ops.field_eval_knn(fc, params, pts, pos, quat, Kn, 10.0, 1.0)
        kk = min(Kn, NF)
        dk, ik = [], []
        for c0 in range(0, P, 512):                                  # exact distances (fp64), brute force
            d = ((pts[c0:c0 + 512, None].double() - pos[None].double()) ** 2).sum(-1).sqrt()
            a, b = d.topk(kk, largest=False)
            dk.append(a); ik.append(b)
        dk, ik = torch.cat(dk), torch.cat(ik)
        w = torch.softmax(-10.0 * dk, -1)
        ref = (w * val[ik].double()).sum(-1)
        ref = torch.where(dk[:, 0] < 1.0, ref, torch.ones_like(ref))
        err = (out[:, 0].double() - ref).abs()
        err[(dk[:, 0] - 1.0).abs() < 1e-5] = 0                       # a point on a sphere may fall either way in fp32
        nb = int((err > 2e-4).sum())
        if nb:
            bad += 1
            print(f"iteration {it}: NF={NF} K={Kn} P={P}: {nb} points differ, max {float(err.max()):.3e}")
    print("stress:", "FAILED" if bad else "ok")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40) else 0)
