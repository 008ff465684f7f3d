"""The numpy restatement of the in-kernel noise generator (tests/rng_math.py) against Random123's known-answer vectors for
Philox-4x32-10, and the statistics of its Box-Muller transform.  The GPU side (tests/test_gpu_rng.py) compares the kernels with this
restatement bit for bit (raw words) and to 2e-6 (normals)."""
import numpy as np

from tests import rng_math as R

# Random123 kat_vectors, philox4x32 with 10 rounds: counter words, key words -> output words
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = [int(v) for v in R.philox4x32_10(*ctr, *key)]
        assert got == list(want), ([hex(g) for g in got], [hex(w) for w in want])


def test_counter_layout_is_stateless_in_batch_and_call():
    # a shard that starts at trajectory 100 draws rows 100.. of the whole batch; another call number draws something else
    whole = R.raw_words(seed=12, n=3, first_trajectory=0, B=110, L=8)
    shard = R.raw_words(seed=12, n=3, first_trajectory=100, B=10, L=8)
    assert np.array_equal(whole[100:], shard)
    other = R.raw_words(seed=12, n=4, first_trajectory=0, B=110, L=8)
    assert not np.array_equal(whole, other)
    assert len(np.unique(whole.reshape(-1))) > 0.999 * whole.size


def test_normals_statistics():
    z = R.normals(seed=2026, n=0, first_trajectory=0, B=1 << 16, L=8)
    assert np.isfinite(z).all() and abs(z).max() < 5.8           # 23-bit uniforms: |z| <= sqrt(2 * 24 ln 2) = 5.77
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1.0) < 5e-3
    assert np.abs(np.corrcoef(z.T) - np.eye(8)).max() < 0.02      # the eight latent columns are uncorrelated
    q = np.quantile(z, [0.025, 0.5, 0.975])
    assert abs(q[0] + 1.96) < 0.03 and abs(q[1]) < 0.02 and abs(q[2] - 1.96) < 0.03
