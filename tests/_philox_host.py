"""Host restatement (numpy, uint64 arithmetic) of the jitter generator of the kernels (csrc/ngm_device.h philox_uniform):
the standard Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC11) with
    counter = (idx low 32, idx high 32, stream id, offset low 32),  key = (seed low 32, seed high 32),
first output word >> 8 scaled to [0, 1).  Test infrastructure: tests/test_host_logic.py pins it to the Random123 known-answer
vectors on the CPU, tests/test_gpu_hardening.py compares the in-kernel draws with it bit for bit."""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def philox4x32_10_words(c0, c1, c2, c3, k0, k1):
    """vectorised over uint64 arrays holding 32-bit words; returns the four output words"""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _M, p1 >> np.uint64(32), p1 & _M
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & _M, (k1 + np.uint64(0xBB67AE85)) & _M
    return c0, c1, c2, c3


def philox4x32_10(ctr, key):
    """scalar form for the known-answer vectors: ctr = 4 words, key = 2 words -> 4 words"""
    out = philox4x32_10_words(*[np.array([c], dtype=np.uint64) for c in ctr], key[0], key[1])
    return tuple(int(w[0]) for w in out)


def host_philox_uniform(seed, offset, idx, stream, word=0):
    """`word`: which of the block's four output words (0: every sampler; 1: the offset draw of k_sample_rays_weighted, whose bin
    draw is word 0 of the same block -- ngm_device.h philox_uniform2)"""
    idx = np.asarray(idx, dtype=np.uint64)
    c0 = philox4x32_10_words(idx & _M, idx >> np.uint64(32), np.full_like(idx, np.uint64(stream)),
                             np.full_like(idx, np.uint64(int(offset) & 0xFFFFFFFF)), int(seed) & 0xFFFFFFFF,
                             (int(seed) >> 32) & 0xFFFFFFFF)[word]
    return (c0 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def host_philox_draws(seed, offset, F, R, n_c, n_g):
    """the (F, R, n_c) coarse and (F, R, n_g) guided draws of the fused forward / ngm_sample_rays for (seed, offset): element e
    of global ray f * R + r in stratum `which` is counter ray * n + e of stream `which` (ngm_device.h jitter())"""
    import torch
    ray = np.arange(F * R, dtype=np.uint64)[:, None]
    u_c = host_philox_uniform(seed, offset, ray * np.uint64(n_c) + np.arange(n_c, dtype=np.uint64)[None], 0).reshape(F, R, n_c)
    u_g = (host_philox_uniform(seed, offset, ray * np.uint64(n_g) + np.arange(n_g, dtype=np.uint64)[None], 1).reshape(F, R, n_g)
           if n_g else None)
    return torch.from_numpy(u_c), (torch.from_numpy(u_g) if n_g else None)
