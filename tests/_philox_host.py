"""Host restatement (numpy, uint64 arithmetic) of the jitter generator of the kernels (csrc/ngm_device.h philox_block /
philox_uniform / jitter_fill): the standard Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC11)
with
    counter = (block low 32, block high 32, stream id, offset low 32),  key = (seed low 32, seed high 32);
element idx of a stream is word idx & 3 of block idx >> 2 (round 6: all four words of a block are used; rounds 1-5: block = idx,
word 0), the word >> 8 scaled to [0, 1).  Test infrastructure: tests/test_host_logic.py pins it to the Random123 known-answer
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


def host_philox_uniform(seed, offset, idx, stream, word=None):
    """Element `idx` of stream `stream`.  word=None: the samplers' mapping (block idx >> 2, word idx & 3: ngm_device.h
    philox_uniform / jitter_fill).  word=0 / 1: the weighted-bin sampler's own mapping (ngm_device.h philox_uniform2): block =
    idx, its first word is the bin draw, its second the offset draw."""
    idx = np.asarray(idx, dtype=np.uint64)
    blk = idx if word is not None else idx >> np.uint64(2)
    words = philox4x32_10_words(blk & _M, blk >> np.uint64(32), np.full_like(blk, np.uint64(stream)),
                                np.full_like(blk, np.uint64(int(offset) & 0xFFFFFFFF)), int(seed) & 0xFFFFFFFF,
                                (int(seed) >> 32) & 0xFFFFFFFF)
    if word is not None:
        c = words[word]
    else:
        k = idx & np.uint64(3)
        c = np.where(k == 0, words[0], np.where(k == 1, words[1], np.where(k == 2, words[2], words[3])))
    return (c >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def host_philox_draws(seed, offset, F, R, n_c, n_g):
    """the (F, R, n_c) coarse and (F, R, n_g) guided draws of the fused forward / ngm_sample_rays for (seed, offset): element e
    of global ray f * R + r in stratum `which` is counter ray * n + e of stream `which` (ngm_device.h jitter())"""
    import torch
    ray = np.arange(F * R, dtype=np.uint64)[:, None]
    u_c = host_philox_uniform(seed, offset, ray * np.uint64(n_c) + np.arange(n_c, dtype=np.uint64)[None], 0).reshape(F, R, n_c)
    u_g = (host_philox_uniform(seed, offset, ray * np.uint64(n_g) + np.arange(n_g, dtype=np.uint64)[None], 1).reshape(F, R, n_g)
           if n_g else None)
    return torch.from_numpy(u_c), (torch.from_numpy(u_g) if n_g else None)
