import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def split_prefix(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_sessionfinish(session, exitstatus):
    """-m gpu sessions: write what the oracle comparisons measured (gpu_common.MARGINS / KINK) to gpurun_out/parity_margins.txt"""
    gc = sys.modules.get("gpu_common")
    if gc is None or not (gc.MARGINS or gc.KINK or gc.SWEEPS):
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_margins.txt"), "w") as fh:
        fh.write("# kink_free_draws: rays taken out of the loss before a gradient comparison (test, rays, samples/ray, neutralised, fraction, rounds)\n")
        for k in gc.KINK:
            fh.write(f"kink  {k['test']}  rays={k['rays']} S={k['samples_per_ray']} neutralised={k['neutralised_rays']} "
                     f"frac={k['neutralised_frac']:.5f} narrowed_band_rays={k['narrowed_rays']} min_band={k['min_margin']:.2e} "
                     f"rounds={k['redraw_rounds']}\n")
        fh.write("# gradient comparisons vs the oracle: worst |kernel - oracle| / max |oracle| per tensor, next to the bar it is held to\n")
        worst = {}
        for m in gc.MARGINS:
            fh.write(f"grad  {m['test']}  {m['name']}  err={m['err']:.3e} tol={m['tol']:.1e}"
                     + ("  per_level=" + ",".join(f"{x:.2e}" for x in m["per_level"]) if "per_level" in m else "") + "\n")
            key = (m["name"], m["tol"])
            worst[key] = max(worst.get(key, 0.0), m["err"])
        fh.write("# loss + gradient comparisons against the oracle per test function: instances run / instances that compared every loss "
                 "term and every gradient / instances with an empty loss selection among them (NaN terms, as in the reference)\n")
        for name, rec in sorted(gc.SWEEPS.items()):
            fh.write(f"sweep {name}  compared={rec['compared']}/{rec['run']}  with_empty_selection={rec['empty']}\n")
        fh.write("# worst error per (tensor, bar) over the whole session\n")
        for (name, tol), e in sorted(worst.items()):
            fh.write(f"worst {name}  tol={tol:.1e}  err={e:.3e}  margin={tol / max(e, 1e-30):.1f}x\n")
