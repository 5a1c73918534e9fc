"""threefry2x32 / jax.random restatement: Random123 and JAX known answers."""
import numpy as np

from rsr_mjx_amd import prng


def test_threefry_known_answers(oracle_mod):
    kats = [((0, 0), (0, 0), (0x6B200159, 0x99BA4EFE)),
            ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
            ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0))]
    for key, ctr, want in kats:
        got = oracle_mod.threefry2x32(key, ctr)
        assert tuple(int(x) for x in got) == want
        o0, o1 = prng.threefry2x32(np.array(key, np.uint32), np.uint32(ctr[0]), np.uint32(ctr[1]))
        assert (int(o0), int(o1)) == want


def test_split_and_uniform_match_published_jax_values(oracle_mod):
    # jax.random.split(jax.random.PRNGKey(0)) and jax.random.uniform(jax.random.PRNGKey(0)) (jax docs, threefry default)
    want = np.array([[4146024105, 967050713], [2718843009, 1272950319]], dtype=np.uint32)
    np.testing.assert_array_equal(prng.split(prng.PRNGKey(0), 2), want)
    np.testing.assert_array_equal(oracle_mod.split([0, 0], 2), want)
    assert abs(float(prng.uniform(prng.PRNGKey(0))) - 0.41845703) < 1e-8
    assert abs(float(oracle_mod.uniform([0, 0], 1, 0, 1)[0]) - 0.41845703) < 1e-8


def test_host_prng_equals_oracle_prng(oracle_mod):
    keys = prng.split(prng.PRNGKey(42), 7)
    for k in keys:
        for n in (1, 2, 3, 5, 20, 22):
            np.testing.assert_array_equal(prng.split(k, n), oracle_mod.split(k, n))
            np.testing.assert_array_equal(prng.uniform(k, (n,), -0.01, 0.01), oracle_mod.uniform(k, n, -0.01, 0.01))
    # batched keys == per-key evaluation (vmap over keys is serial evaluation under the non-partitionable impl)
    batched = prng.uniform(keys, (5,), 0.2, 0.9)
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(batched[i], prng.uniform(k, (5,), 0.2, 0.9))
    lo, hi = np.array([0.29, -0.04, 0.82], np.float32), np.array([0.34, 0.01, 0.82], np.float32)
    u = prng.uniform(keys[0], (3,), lo, hi)
    assert u[2] == np.float32(0.82) and lo[0] <= u[0] <= hi[0]
