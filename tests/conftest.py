import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def ctx():
    """One plk_ctx on cuda:0.  Fails (never skips) when the library or the GPU is missing."""
    from plonk_amd.hip import Context
    c = Context(0)
    yield c
    c.close()


def independent_polys(orc, tables):
    """a, b, c, q_add, q_mul, s_id, s_sigma of src/main.rs:79-84,187-191 built by the CPU oracle alone, so that a proof or an
    interpolation of the GPU is never checked against the GPU's own output: the domain elements w^i come from the oracle's forward
    transform of the polynomial X, the permutation values are gathered from them with numpy (src/poly_utils.rs:18-25), and the four
    interpolations are the oracle's inverse transforms."""
    import numpy as np
    from plonk_amd import synth
    n = tables["n"]
    n4 = 1 << (3 * n - 1).bit_length()
    x_poly = np.zeros((n4, 4), dtype=np.uint64)
    x_poly[1] = synth.to_mont([1])[0]
    domain = orc.ntt(x_poly)                                                   # domain.element(i) = w^i
    sigma = np.concatenate([np.asarray(tables["sigma"], dtype=np.int64), np.arange(3 * n, n4, dtype=np.int64)])
    # s_id = ifft(w^i) is the polynomial X (src/poly_utils.rs:24,27); above 2^22 points the closed form saves half a minute of CPU time
    s_id = synth.trim(orc.ntt(domain, inverse=True)) if n4 <= (1 << 22) else synth.to_mont([0, 1])
    s_sigma = synth.trim(orc.ntt(domain[sigma], inverse=True))
    q_add = synth.trim(orc.ntt(tables["q_add_vals"], inverse=True))
    q_mul = synth.trim(orc.ntt(tables["q_mul_vals"], inverse=True))
    return [synth.trim(tables["a"]), synth.trim(tables["b"]), synth.trim(tables["c"]), q_add, q_mul, s_id, s_sigma]


def assert_polys_equal(got, want):
    """byte equality of the seven prover polynomials, and s_id = X (src/poly_utils.rs:24,27)"""
    import numpy as np
    from plonk_amd import synth
    names = ["a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma"]
    for name, g, w in zip(names, got, want):
        assert g.shape == w.shape and np.array_equal(g, w), name
    assert synth.from_mont(got[5]) == [0, 1]
