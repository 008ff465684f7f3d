"""numpy restatement of the engine's in-kernel noise generator (csrc/slode_common.h: philox4x32_10, slode_rng_normal) -- test
infrastructure.  Philox-4x32-10 is Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); the known-answer
vectors in tests/test_rng_cpu.py are the ones Random123 ships for it."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays (or scalars) of uint32 counter / key words -> four arrays of uint32 output words."""
    c0, c1, c2, c3, k0, k1 = [np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3, k0, k1)]
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                      # 32 x 32 -> 64 bit products (no overflow in uint64)
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return [v.astype(np.uint32) for v in (c0, c1, c2, c3)]


def raw_words(seed, n, first_trajectory, B, L):
    """[B, ceil(L/4), 4] uint32: block (b, j) = philox(counter = [b + first_trajectory | j | n lo | n hi], key = seed lo, hi)."""
    nb = (L + 3) // 4
    b = (np.arange(B, dtype=np.uint64) + np.uint64(first_trajectory))[:, None] + np.zeros((1, nb), dtype=np.uint64)
    j = np.zeros((B, 1), dtype=np.uint64) + np.arange(nb, dtype=np.uint64)[None, :]
    w = philox4x32_10(b, j, np.uint64(n) & MASK, np.uint64(n) >> np.uint64(32), np.uint64(seed) & MASK, np.uint64(seed) >> np.uint64(32))
    return np.stack(w, axis=-1)


def normals(seed, n, first_trajectory, B, L):
    """[B, L] float64: word pair (0, 1) of block l >> 2 serves l & 2 == 0, pair (2, 3) the rest; u = ((x >> 9) + 0.5) 2^-23,
    radius sqrt(-2 ln u_a), angle 2 pi u_b, cosine for even l, sine for odd l."""
    w = raw_words(seed, n, first_trajectory, B, L).astype(np.uint64)
    l = np.arange(L)
    blk, hi, odd = l >> 2, (l & 2) != 0, (l & 1) != 0
    xa = np.where(hi[None, :], w[:, blk, 2], w[:, blk, 0])
    xb = np.where(hi[None, :], w[:, blk, 3], w[:, blk, 1])
    ua = ((xa >> np.uint64(9)).astype(np.float64) + 0.5) * 2.0 ** -23
    ub = ((xb >> np.uint64(9)).astype(np.float64) + 0.5) * 2.0 ** -23
    r = np.sqrt(-2.0 * np.log(ua))
    return r * np.where(odd[None, :], np.sin(2 * np.pi * ub), np.cos(2 * np.pi * ub))
