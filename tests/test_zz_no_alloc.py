"""plk_poly_eval_dev_static (the entry point that allocates nothing) against plk_poly_eval_dev, through the C ABI."""
import numpy as np
import pytest

from plonk_amd import hip, synth

pytestmark = pytest.mark.gpu


def rand_fr(orc, n, seed):
    rng = np.random.default_rng(seed)
    return orc.fr_from_ints([int.from_bytes(rng.bytes(40), "little") for _ in range(n)]) if n else np.zeros((0, 4), np.uint64)


def test_static_evaluation_matches_the_allocating_one(ctx, orc):
    """Two calls in a row on the same GPU, the second with a smaller input: the fixed scratch array keeps the first call's block
    sums, which the second must not pick up.  Bit-exact against plk_poly_eval_dev and the CPU oracle."""
    x = rand_fr(orc, 1, 99)[0]
    d_out = ctx.malloc(32)
    ctx.upload_into(d_out, np.full((4,), 0xFFFFFFFFFFFFFFFF, np.uint64))
    for n, seed in ((3 * 8192 + 77, 5), (8192 + 5, 6), (1, 7), (0, 8)):   # several blocks, then fewer, then one coefficient, then none
        a = rand_fr(orc, n, seed)
        d = ctx.upload(a)
        want = ctx.poly_eval_dev(d, n, x)
        ctx.poly_eval_dev_static(d, n, x, d_out)
        got = ctx.download(d_out, (4,))
        assert np.array_equal(got, want), n
        assert np.array_equal(got, orc.poly_eval(a, x) if n else np.zeros(4, np.uint64)), n
        ctx.free(d)
    ctx.free(d_out)


def test_static_evaluation_full_size(ctx, orc):
    """2^20 and then 2^18 coefficients (the sizes of a 2^20-gate proof's polynomials), against the allocating entry point"""
    x = rand_fr(orc, 1, 3)[0]
    d_out = ctx.malloc(32)
    rng = np.random.default_rng(11)
    for log_n in (20, 18):
        a = rng.integers(0, 1 << 62, size=(1 << log_n, 4), dtype=np.uint64)   # limbs below 2^62: canonical values below the modulus
        a[:, 3] &= (1 << 60) - 1
        d = ctx.upload(a)
        want = ctx.poly_eval_dev(d, 1 << log_n, x)
        ctx.poly_eval_dev_static(d, 1 << log_n, x, d_out)
        assert np.array_equal(ctx.download(d_out, (4,)), want), log_n
        ctx.free(d)
    ctx.free(d_out)


def test_static_evaluation_rejects_lengths_beyond_its_scratch(ctx):
    d = ctx.malloc(64)
    with pytest.raises(hip.PlonkHipError) as e:
        ctx.poly_eval_dev_static(d, (1 << 26) + 1, np.zeros(4, np.uint64), d)
    assert e.value.status == "PLK_ERR_INVALID_ARG" and "PLK_POLY_EVAL_STATIC_MAX_LEN" in str(e.value)
    ctx.free(d)
