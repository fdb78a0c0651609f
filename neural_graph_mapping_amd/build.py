"""Build the gfx950 C-ABI library (libngm_hip.so) in-tree with hipcc.

    python -m neural_graph_mapping_amd.build [--fast] [--force] [--prune]

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached under csrc/_obj (keyed by a hash
of the sources + flags); the shared library lands in neural_graph_mapping_amd/lib/ and travels to
the GPU box with the repo snapshot.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libngm_hip.so")
SOURCES = ["ngm_api.hip", "ngm_field_fwd.hip", "ngm_field_bwd.hip", "ngm_field_bwd16.hip", "ngm_field_bwd16s.hip", "ngm_field_bwd_b3.hip", "ngm_hash_bwd.hip", "ngm_target.hip", "ngm_composite.hip",
           "ngm_knn.hip", "ngm_mesh.hip", "ngm_peer.hip"]
def _headers():
    """Every header any source may include: all of csrc/*.h + the public C ABI header (the digest of an
    object covers all of them, so editing any header invalidates the cached objects)."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/ngm_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(src, flags):
    h = hashlib.sha256()
    for f in [src] + _headers():
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:16]


def _compile(src, flags):
    obj = os.path.join(OBJ, src.replace(".hip", "") + "-" + _digest(src, flags) + ".o")
    if not os.path.exists(obj):
        cmd = [_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj


def build(fast=False, force=False, verbose=True, out=None, prune=False):
    """out: file name of a VARIANT library (developer experiments: `NGM_HIPCC_EXTRA="-DX" ... --out=libngm_x.so`,
    loaded by the tools through NGM_LIB_PATH); the product library is always lib/libngm_hip.so."""
    global LIB
    LIB = os.path.join(LIBDIR, out) if out else os.path.join(LIBDIR, "libngm_hip.so")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    import fcntl
    lock = open(os.path.join(OBJ, ".lock"), "w")          # one build at a time per object directory (compile + link + prune)
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        return _build_locked(fast, force, verbose, out, prune)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(fast, force, verbose, out, prune):
    global LIB
    flags = FLAGS + (["-DNGM_FAST_BUILD"] if fast else [])
    flags += os.environ.get("NGM_HIPCC_EXTRA", "").split()      # e.g. -DNGM_PHASE_TIMING (debug builds)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, flags), SOURCES))
    stamp = os.path.join(OBJ, "link.stamp" if not out else "link." + out + ".stamp")
    key = " ".join(objs)
    if force or not os.path.exists(LIB) or not os.path.exists(stamp) or open(stamp).read() != key:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        with open(stamp, "w") as fh:
            fh.write(key)
    if prune and not out and not os.environ.get("NGM_HIPCC_EXTRA"):
        # `--prune` only (explicit): objects of earlier source states are dead weight on the way to the GPU box, but deleting them
        # by default made alternating fast / full builds evict each other and could take objects away from a build running in
        # another process between its compile and its link.  Objects younger than ten minutes are never touched.
        import time
        keep = {os.path.basename(o) for o in objs}
        for f in os.listdir(OBJ):
            fp = os.path.join(OBJ, f)
            if f.endswith(".o") and f.startswith("ngm_") and f not in keep and time.time() - os.path.getmtime(fp) > 600:
                os.remove(fp)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    outs = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--out=")]
    build(fast="--fast" in sys.argv, force="--force" in sys.argv, out=outs[0] if outs else None, prune="--prune" in sys.argv)
