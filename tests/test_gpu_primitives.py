"""HIP library vs the CPU oracle, through the C ABI, on seeded inputs (bit-exact: all integer work)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pymodel as pm

pytestmark = pytest.mark.gpu

TAU, GSCAL = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321


def rand_fr(orc, n, seed):
    rng = pm.SplitMix64(seed)
    return orc.fr_from_ints([rng.next_fr() for _ in range(n)])


@pytest.mark.parametrize("field", ["fr", "fq"])
@pytest.mark.parametrize("op", ["add", "sub", "mul"])
def test_field_ops(ctx, orc, field, op):
    rng = pm.SplitMix64(11)
    p = pm.R_MOD if field == "fr" else pm.Q_MOD
    xs = [rng.next_fr() for _ in range(500)] + [0, 1, p - 1, p - 1, 0, 2]
    ys = [rng.next_fr() for _ in range(500)] + [0, p - 1, p - 1, 1, 5, p - 2]
    conv = orc.fr_from_ints if field == "fr" else orc.fq_from_ints
    a, b = conv(xs), conv(ys)
    got = ctx.field_op(field, op, a, b)
    want = (orc.fr_op if field == "fr" else orc.fq_op)(op, a, b)
    assert np.array_equal(got, want)


@pytest.fixture(scope="module")
def srs(ctx, orc):
    n = 1 << 12
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    ctx.srs_generate(tau, gs, n + 1)
    g = ctx.srs_read(0, 0, n + 1)
    gg = ctx.srs_read(1, 0, 3)
    return dict(n=n, g=g, gg=gg, tau=tau, gs=gs)


def test_srs_generate_matches_oracle(ctx, orc, srs):
    one = orc.fr_from_ints([1])[0]
    want = orc.srs_powers(srs["tau"], one, 0, 64)
    assert np.array_equal(srs["g"][:64], want)
    want_hi = orc.srs_powers(srs["tau"], one, srs["n"] - 3, 4)
    assert np.array_equal(srs["g"][srs["n"] - 3: srs["n"] + 1], want_hi)
    assert np.array_equal(srs["gg"], orc.srs_powers(srs["tau"], srs["gs"], 0, 3))


def test_srs_load_roundtrip(orc):
    from plonk_amd.hip import Context
    c2 = Context(0)
    one = orc.fr_from_ints([1])[0]; tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    g = orc.srs_powers(tau, one, 0, 40); gg = orc.srs_powers(tau, gs, 0, 3)
    c2.srs_load(g, gg)
    assert np.array_equal(c2.srs_read(0, 0, 40), g)
    assert np.array_equal(c2.srs_read(1, 0, 3), gg)
    sc = rand_fr(orc, 40, 5)
    got, inf = c2.msm(sc)
    want, winf = orc.msm(sc, g)
    assert inf == winf == 0 and np.array_equal(got, want)
    c2.close()


@pytest.mark.parametrize("k", [0, 1, 2, 3, 17, 64, 1000, 4096, 4097])
def test_msm_random(ctx, orc, srs, k):
    sc = rand_fr(orc, k, 100 + k)
    got, inf = ctx.msm(sc)
    want, winf = orc.msm(sc, srs["g"][:k])
    assert inf == winf
    assert np.array_equal(got, want)


def test_msm_edge_scalars(ctx, orc, srs):
    r = pm.R_MOD
    vals = [0, 1, 2, r - 1, r - 2, 1 << 15, (1 << 15) + 1, (1 << 16) - 1, 1 << 16, (1 << 253), 3, 7, 14, 19, 57, 0x8000800080008000, 0xFFFFFFFFFFFFFFFF]
    sc = orc.fr_from_ints(vals)
    got, inf = ctx.msm(sc, base_offset=5)
    want, winf = orc.msm(sc, srs["g"][5:5 + len(vals)])
    assert inf == winf == 0 and np.array_equal(got, want)
    # all scalars equal (one heavy bucket per window) and cancellation to the identity
    sc = orc.fr_from_ints([12345] * 300)
    got, inf = ctx.msm(sc)
    want, winf = orc.msm(sc, srs["g"][:300])
    assert inf == winf == 0 and np.array_equal(got, want)
    zero = orc.fr_from_ints([0] * 10)
    got, inf = ctx.msm(zero)
    assert inf == 1


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_ntt_matches_oracle(ctx, orc, log_n):
    """every plan of the transform up to 2^22 points (the s_sigma transform of a 2^20-gate proof): the single-pass kernel below 2^11, then two and three passes with every
    mix of pass radices (2^5 .. 2^8) and of butterfly-stage groupings inside a pass"""
    a = rand_fr(orc, 1 << log_n, 7 + log_n)
    assert np.array_equal(ctx.ntt(a), orc.ntt(a))
    assert np.array_equal(ctx.ntt(a, inverse=True), orc.ntt(a, inverse=True))
    g = orc.fr_from_ints([5])[0]
    assert np.array_equal(ctx.ntt(a, coset=g), orc.ntt(a, coset=g))
    assert np.array_equal(ctx.ntt(a, inverse=True, coset=g), orc.ntt(a, inverse=True, coset=g))


@pytest.mark.parametrize("length", [0, 1, 2, 31, 32, 33, 1024, 1025, 5000])
def test_poly_eval(ctx, orc, length):
    c = rand_fr(orc, length, 3 + length)
    x = rand_fr(orc, 1, 99)[0]
    got = ctx.poly_eval(c, x)
    want = orc.poly_eval(c, x) if length else np.zeros(4, dtype=np.uint64)
    assert np.array_equal(got, want)


def _circuit(orc, log_n, seed):
    variables, gates = pm.synthetic_circuit(log_n, seed)
    inp = pm.prepare_inputs(variables, gates)
    return inp


@pytest.mark.parametrize("log_n", [2, 3, 5, 8])
def test_grand_product_and_quotient(ctx, orc, log_n):
    inp = _circuit(orc, log_n, 21 + log_n)
    n = inp["n"]
    w = orc.fr_from_ints(inp["witness_flat"]); sigma = np.array(inp["sigma"], dtype=np.uint64)
    beta, gamma = rand_fr(orc, 2, 77)
    zv, zc = orc.grand_product(w, sigma, beta, gamma)
    got = ctx.grand_product(w, sigma, beta, gamma)
    assert np.array_equal(got, zc)
    a, b, c, qa, qm = (orc.fr_from_ints(col) for col in inp["cols"])
    t, tlen = ctx.quotient(a, b, c, qa, qm, zv, sigma, beta, gamma)
    assert tlen == 0 and not t.any()
    assert orc.quotient(a, b, c, qa, qm, zv, sigma, beta, gamma) == 0
    # a broken witness must be rejected like the reference's assert
    from plonk_amd.hip import PlonkHipError
    bad = c.copy(); bad[1] = a[0]
    with pytest.raises(PlonkHipError) as e:
        ctx.quotient(a, b, bad, qa, qm, zv, sigma, beta, gamma)
    assert e.value.status == "PLK_ERR_NOT_DIVISIBLE"
    assert orc.quotient(a, b, bad, qa, qm, zv, sigma, beta, gamma) == 1


@pytest.mark.parametrize("log_n", [11, 12, 15])
def test_grand_product_of_random_rows(ctx, orc, log_n):
    """One, two and sixteen scan blocks (2048 rows each): any rows and any permutation have a grand product, no circuit needed.
    Values and coefficients against the oracle, through the waiting call and through the values-only enqueue variant."""
    n = 1 << log_n
    rng = np.random.default_rng(300 + log_n)
    w = rand_fr(orc, 3 * 257, 40 + log_n)[rng.integers(0, 3 * 257, 3 * n)]
    sigma = rng.permutation(3 * n).astype(np.uint64)
    beta, gamma = rand_fr(orc, 2, 60 + log_n)
    zv, zc = orc.grand_product(w, sigma, beta, gamma)
    d_w, d_s = ctx.upload(w), ctx.upload(sigma)
    d_v, d_c = ctx.malloc(n * 32), ctx.malloc(n * 32)
    ctx.grand_product_values_dev(d_w, d_s, log_n, beta, gamma, d_v, d_c)
    assert np.array_equal(ctx.download(d_v, (n, 4)), zv) and np.array_equal(ctx.download(d_c, (n, 4)), zc)
    ctx.upload_into(d_v, np.zeros((n, 4), np.uint64))
    ctx.grand_product_dev_enqueue(d_w, d_s, log_n, beta, gamma, d_v, None)
    ctx.sync()
    assert np.array_equal(ctx.download(d_v, (n, 4)), zv)
    for d in (d_w, d_s, d_v, d_c):
        ctx.free(d)


def test_grand_product_values_and_ticket_batches(ctx, orc, srs):
    """plk_grand_product_values_dev hands out z on H beside the coefficients; tickets of one batch may be ended in any order,
    with or without plk_kzg_submit."""
    inp = _circuit(orc, 6, 17)
    n = inp["n"]
    w = orc.fr_from_ints(inp["witness_flat"]); sigma = np.array(inp["sigma"], dtype=np.uint64)
    beta, gamma = rand_fr(orc, 2, 5)
    zv, zc = orc.grand_product(w, sigma, beta, gamma)
    d_w, d_s = ctx.upload(w), ctx.upload(sigma)
    d_v, d_c = ctx.malloc(n * 32), ctx.malloc(n * 32)
    ctx.grand_product_values_dev(d_w, d_s, 6, beta, gamma, d_v, d_c)
    assert np.array_equal(ctx.download(d_v, (n, 4)), zv) and np.array_equal(ctx.download(d_c, (n, 4)), zc)
    # the variant that does not wait: a commitment begun right after it is ordered behind the interpolation
    d_v2, d_c2 = ctx.malloc(n * 32), ctx.malloc(n * 32)
    ctx.grand_product_dev_enqueue(d_w, d_s, 6, beta, gamma, d_v2, d_c2)
    z_blind = rand_fr(orc, 3, 949)
    z_ticket = ctx.commit_begin(d_c2, n, z_blind)
    ctx.sync()
    assert np.array_equal(ctx.download(d_v2, (n, 4)), zv) and np.array_equal(ctx.download(d_c2, (n, 4)), zc)
    got, inf = ctx.commit_end(z_ticket)
    want_z = orc.kzg_commit(srs["g"], srs["gg"], zc, z_blind)
    assert inf == want_z[1] and np.array_equal(got, want_z[0])
    polys = [rand_fr(orc, k, 900 + k) for k in (64, 0, 1, 4096, 100)]
    blinds = [rand_fr(orc, 3, 950 + i) for i in range(len(polys))]
    want = [orc.kzg_commit(srs["g"], srs["gg"], p, b) for p, b in zip(polys, blinds)]
    dev = [ctx.upload(p) for p in polys]
    for use_submit in (False, True):
        tickets = [ctx.commit_begin(d, len(p), b) for d, p, b in zip(dev, polys, blinds)]
        if use_submit:
            ctx.submit()
        for i in (3, 0, 4, 1, 2):
            got, inf = ctx.commit_end(tickets[i])
            assert inf == want[i][1] and np.array_equal(got, want[i][0])
    for d in dev + [d_w, d_s, d_v, d_c, d_v2, d_c2]:
        ctx.free(d)


def test_commitments_see_compute_work_enqueued_between_begins(ctx, orc, srs):
    """Jobs of one batch alternate between two front streams, only one of which carries the compute work: a transform enqueued
    between two *_begin calls must be finished before the commitment that reads its output starts, whichever stream that
    commitment's sort lands on."""
    log_n = 11
    polys = [rand_fr(orc, 1 << log_n, 700 + i) for i in range(5)]
    blinds = [rand_fr(orc, 3, 720 + i) for i in range(5)]
    devs = [ctx.upload(p) for p in polys]
    tickets = []
    for d, b in zip(devs, blinds):
        ctx.ntt_dev_enqueue(d, log_n)
        tickets.append(ctx.commit_begin(d, 1 << log_n, b))
    for t, p, b in zip(tickets, polys, blinds):
        got, inf = ctx.commit_end(t)
        want = orc.kzg_commit(srs["g"], srs["gg"], orc.ntt(p), b)
        assert inf == want[1] and np.array_equal(got, want[0])
    for d in devs:
        ctx.free(d)


@pytest.mark.parametrize("log_n,coset", [(0, None), (1, None), (2, None), (3, 7), (5, None), (9, None), (12, 7)])
def test_quotient_on_the_extended_coset(ctx, orc, log_n, coset):
    """plk_quotient_coset against the oracle, which divides in coefficient form and never touches a coset."""
    inp = _circuit(orc, log_n, 31 + log_n)
    w = orc.fr_from_ints(inp["witness_flat"]); sigma = np.array(inp["sigma"], dtype=np.uint64)
    beta, gamma, alpha = rand_fr(orc, 3, 78 + log_n)
    zv, _ = orc.grand_product(w, sigma, beta, gamma)
    a, b, c, qa, qm = (orc.fr_from_ints(col) for col in inp["cols"])
    g = orc.fr_from_ints([coset])[0] if coset is not None else None
    got = ctx.quotient_coset(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, coset=g)
    rc, want = orc.quotient_coset(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha)
    assert rc == 0 and np.array_equal(got, want)
    if log_n >= 2:
        assert got.any()                                   # unlike the reference's t, this quotient is not the zero polynomial
        from plonk_amd.hip import PlonkHipError
        bad = c.copy(); bad[1] = a[0]
        with pytest.raises(PlonkHipError) as e:
            ctx.quotient_coset(a, b, bad, qa, qm, zv, sigma, beta, gamma, alpha, coset=g)
        assert e.value.status == "PLK_ERR_NOT_DIVISIBLE"
        assert orc.quotient_coset(a, b, bad, qa, qm, zv, sigma, beta, gamma, alpha)[0] == 1


@pytest.mark.parametrize("log_n", [2, 5, 10])
def test_quotient_with_public_inputs(ctx, orc, log_n):
    """Circuit::build_public_input_poly inside build_quotient_polynomial (src/circuit.rs:410-439, :458, :469): the term
    alpha * (a[row] - pi) is zero exactly when the row's first wire carries the public value; any other value trips the
    reference's divisibility assert (:487).  Same verdicts as the oracle for matching, mismatching and overwritten entries."""
    from plonk_amd.hip import PlonkHipError
    inp = _circuit(orc, log_n, 17 + log_n)
    n = inp["n"]
    w = orc.fr_from_ints(inp["witness_flat"]); sigma = np.array(inp["sigma"], dtype=np.uint64)
    beta, gamma, alpha = rand_fr(orc, 3, 91 + log_n)
    zv, _ = orc.grand_product(w, sigma, beta, gamma)
    a, b, c, qa, qm = (orc.fr_from_ints(col) for col in inp["cols"])
    rows = np.array([0, n - 1, n // 2], dtype=np.uint64)
    good = a[rows.astype(np.int64)]
    # public values that equal the wires they are compared with: the quotient is the zero polynomial, as without public inputs
    assert orc.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows, good) == 0
    t, tlen = ctx.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows, good)
    assert tlen == 0 and not t.any()
    # one value off by one: not divisible, on both sides
    bad = good.copy(); bad[1] = orc.fr_op("add", bad[1:2], orc.fr_from_ints([1]))[0]
    assert orc.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows, bad) == 1
    with pytest.raises(PlonkHipError) as e:
        ctx.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows, bad)
    assert e.value.status == "PLK_ERR_NOT_DIVISIBLE"
    # the wrong entry is replaced by a later, matching entry for the same row (the reference assigns constraint[row])
    rows2 = np.concatenate([rows, rows[1:2]]); vals2 = np.concatenate([bad, good[1:2]])
    assert orc.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows2, vals2) == 0
    t, tlen = ctx.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows2, vals2)
    assert tlen == 0 and not t.any()
    # ... and the other way round
    rows3 = np.concatenate([rows, rows[1:2]]); vals3 = np.concatenate([good, bad[1:2]])
    assert orc.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows3, vals3) == 1
    with pytest.raises(PlonkHipError) as e:
        ctx.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, rows3, vals3)
    assert e.value.status == "PLK_ERR_NOT_DIVISIBLE"
    # alpha = 0 switches the term off whatever the values are
    zero = orc.fr_from_ints([0])[0]
    assert orc.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, zero, rows, bad) == 0
    assert ctx.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, zero, rows, bad)[1] == 0
    # a row outside the domain is the reference's index panic
    assert orc.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, [n], good[:1]) == 2
    with pytest.raises(PlonkHipError) as e:
        ctx.quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, [n], good[:1])
    assert e.value.status == "PLK_ERR_INDEX_OUT_OF_RANGE"


def test_interpolate_permutation(ctx, orc):
    inp = _circuit(orc, 4, 5)
    log_n4 = inp["n4"].bit_length() - 1
    sid, ssg = ctx.interpolate_permutation(np.array(inp["sigma"], dtype=np.uint64), log_n4)
    assert pm.trim(orc.fr_to_ints(sid)) == inp["s_id"] == [0, 1]
    assert pm.trim(orc.fr_to_ints(ssg)) == inp["s_sigma"]
    qa = orc.fr_from_ints(inp["cols"][3])
    assert pm.trim(orc.fr_to_ints(ctx.interpolate_selector(qa))) == inp["q_add"]


@pytest.mark.parametrize("length", [0, 1, 2, 3, 100, 4096])
def test_kzg_commit_open(ctx, orc, srs, length):
    coeffs = rand_fr(orc, length, 40 + length)
    blind = rand_fr(orc, 3, 41 + length)
    zeta, chal = rand_fr(orc, 2, 42 + length)
    chal = orc.fr_from_ints([orc.fr_to_ints(chal)[0] & ((1 << 128) - 1)])[0]
    got, inf = ctx.kzg_commit(coeffs, blind)
    want, winf = orc.kzg_commit(srs["g"], srs["gg"], coeffs, blind)
    assert inf == winf and np.array_equal(got, want)
    w, winf2, rv, ev = ctx.kzg_open(coeffs, blind, zeta, chal)
    ow, oinf, orv = orc.kzg_open(srs["g"], srs["gg"], coeffs, blind, zeta, chal)
    assert winf2 == oinf and np.array_equal(w, ow)
    assert np.array_equal(rv, orv)
    want_ev = orc.poly_eval(coeffs, zeta) if length else np.zeros(4, dtype=np.uint64)
    assert np.array_equal(ev, want_ev)


@pytest.mark.parametrize("short_max,mid_max,bits", [("0", "0", 17), ("0", "1000000", 16), ("1000000", "0", 13)])
def test_msm_edge_cases_on_every_window_plan(orc, monkeypatch, short_max, mid_max, bits):
    """The library has three window plans: 17-bit windows (2^16 buckets, 15 digits per scalar), 16-bit windows (2^15 buckets, 16
    digits) for MSMs of at most PLK_MSM_MID_MAX scalars and 13-bit windows (2^12 buckets, 20 digits) for MSMs of at most
    PLK_MSM_SHORT_MAX scalars.  The settings are read when the SRS tables are built, so each plan can be forced for a whole
    context.  All must agree with the oracle on the scalars that exercise the recoding (0, 1, r - 1, digits 2^(C-1), 2^(C-1) + 1,
    all-ones windows, carries through every window), on repeated bases (doubling and cancellation branches), on ragged lengths,
    and on hiding commitments / openings begun singly and as batches."""
    from plonk_amd.hip import Context
    monkeypatch.setenv("PLK_MSM_SHORT_MAX", short_max)
    monkeypatch.setenv("PLK_MSM_MID_MAX", mid_max)
    c = Context(0)
    n = 5000
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    c.srs_generate(tau, gs, n + 1)
    assert c.msm_plan(n) == (bits, -(-254 // bits), 1 << (bits - 1)) and c.msm_plan(1, base_offset=n) == c.msm_plan(n)
    one = orc.fr_from_ints([1])[0]
    g = orc.srs_powers(tau, one, 0, n + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    assert np.array_equal(c.srs_read(0, 0, n + 1), g) and np.array_equal(c.srs_read(1, 0, 3), gg)
    r = pm.R_MOD
    edge = [0, 1, 2, r - 1, r - 2, (1 << 12), (1 << 12) + 1, (1 << 12) - 1, (1 << 13) - 1, 1 << 13, (1 << 15), (1 << 15) + 1, (1 << 16) - 1, 1 << 16,
            (1 << 253) + 5, int("1" * 254, 2) % r, sum(1 << (13 * j + 12) for j in range(19)), sum(1 << (16 * j + 15) for j in range(15)),
            sum(((1 << 12) + 1) << (13 * j) for j in range(19)) % r, sum(((1 << 15) + 1) << (16 * j) for j in range(15)) % r,
            1 << 17, (1 << 17) - 1, (1 << 16) + 1, sum(1 << (17 * j + 16) for j in range(14)), sum(((1 << 16) + 1) << (17 * j) for j in range(14)) % r,
            sum(((1 << 17) - 1) << (17 * j) for j in range(14)), (((1 << 16) - 1) << 238) % r, r - (1 << 16), r - (1 << 16) - 1]
    sc = orc.fr_from_ints(edge)
    for k in (len(edge), 1, 5):
        for off in (0, 7, n - len(edge)):
            got, inf = c.msm(sc[:k], base_offset=off)
            want, winf = orc.msm(sc[:k], g[off:off + k])
            assert inf == winf and np.array_equal(got, want), (k, off)
    # the same base many times: sum of +-P with doublings and cancellations inside one bucket
    for pattern in ([5] * 64, [7, r - 7] * 32, [1 << 12] * 33, [r - (1 << 15)] * 17 + [1 << 15] * 17):
        scal = orc.fr_from_ints(pattern)
        got, inf = c.msm(scal, base_offset=3)
        want, winf = orc.msm(scal, g[3:3 + len(pattern)])
        assert inf == winf and np.array_equal(got, want), pattern[:2]
    rep = np.repeat(g[11:12], 40, axis=0)
    c2 = Context(0)
    c2.srs_load(rep, gg)
    for pattern in ([3] * 40, [9, r - 9] * 20, list(range(1, 41))):
        scal = orc.fr_from_ints(pattern)
        got, inf = c2.msm(scal)
        want, winf = orc.msm(scal, rep[:len(pattern)])
        assert inf == winf and np.array_equal(got, want), pattern[:2]
    c2.close()
    for k in (0, 1, 2, 3, 17, 64, 1000, 4096, 4097, n):
        s_ = rand_fr(orc, k, 6000 + k)
        got, inf = c.msm(s_)
        want, winf = orc.msm(s_, g[:k])
        assert inf == winf and np.array_equal(got, want), k
    # commitments and openings, single and batched, ragged lengths
    zeta, chal = rand_fr(orc, 2, 4321)
    chal = orc.fr_from_ints([orc.fr_to_ints(chal)[0] & ((1 << 128) - 1)])[0]
    lengths = [0, 1, 2, 3, 100, 2049, n]
    polys = [rand_fr(orc, k, 7000 + k) for k in lengths]
    blinds = np.stack([rand_fr(orc, 3, 7100 + k) for k in lengths])
    devs = [c.upload(p) for p in polys]
    tickets = c.commit_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12))
    c.submit()
    for i in (6, 0, 3, 1, 5, 2, 4):
        got, inf = c.commit_end(tickets[i])
        want, winf = orc.kzg_commit(g, gg, polys[i], blinds[i])
        assert inf == winf and np.array_equal(got, want), lengths[i]
        single = c.kzg_commit(polys[i], blinds[i])
        assert single[1] == winf and np.array_equal(single[0], want), lengths[i]
    tickets = c.open_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12), zeta, chal)
    for i in range(len(lengths)):
        w, winf, rv, ev = c.open_end(tickets[i])
        ow, oinf, orv = orc.kzg_open(g, gg, polys[i], blinds[i], zeta, chal)
        assert winf == oinf and np.array_equal(w, ow) and np.array_equal(rv, orv), lengths[i]
        assert np.array_equal(ev, orc.poly_eval(polys[i], zeta) if lengths[i] else np.zeros(4, dtype=np.uint64)), lengths[i]
    # all tickets of a round ended with one call (plk_kzg_end_many: the host-side ends of the MSMs run on several threads), as affine
    # points and as the partial sums a sharded prover exchanges
    tickets = c.commit_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12))
    pts, inf, _, _ = c.end_many(tickets[::-1])
    for j, i in enumerate(reversed(range(len(lengths)))):
        want, winf = orc.kzg_commit(g, gg, polys[i], blinds[i])
        assert inf[j] == winf and np.array_equal(pts[j, :8], want), lengths[i]
    tickets = c.open_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12), zeta, chal)
    parts, _, rvs, evs = c.end_many(tickets, affine=False)
    xy, inf = c.sum_partials_many(parts.reshape(1, len(lengths), 16))
    for i in range(len(lengths)):
        ow, oinf, orv = orc.kzg_open(g, gg, polys[i], blinds[i], zeta, chal)
        assert inf[i] == oinf and np.array_equal(xy[i], ow) and np.array_equal(rvs[i], orv), lengths[i]
        assert np.array_equal(evs[i], orc.poly_eval(polys[i], zeta) if lengths[i] else np.zeros(4, dtype=np.uint64)), lengths[i]
    for d in devs:
        c.free(d)
    c.close()


def test_open_batch_matches_single_openings_and_the_oracle(orc):
    """plk_kzg_open_begin_batch (the nine openings of src/prover.rs:295-303 as one call): same points, random_v and evaluations as
    plk_kzg_open on each polynomial and as the oracle's long division + MSM, for ragged lengths that include the empty polynomial,
    constants, both sides of the division's block boundaries and, in one batch, polynomials on both sides of the change of block
    size at 2^21 coefficients; ended in an order different from the one they were begun in; twice in a row on one context."""
    from plonk_amd.hip import Context, PlonkHipError
    c = Context(0)
    n = (1 << 21) + 4096
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    c.srs_generate(tau, gs, n + 1)
    one = orc.fr_from_ints([1])[0]
    small = 1 << 14
    g = orc.srs_powers(tau, one, 0, small + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    assert np.array_equal(c.srs_read(0, 0, small + 1), g) and np.array_equal(c.srs_read(1, 0, 3), gg)
    zeta, chal = rand_fr(orc, 2, 1234)
    chal = orc.fr_from_ints([orc.fr_to_ints(chal)[0] & ((1 << 128) - 1)])[0]
    lengths = [0, 1, 2, 3, 2047, 2048, 2049, 4097, 10000, small]
    polys = [rand_fr(orc, k, 2000 + k) for k in lengths]
    blinds = np.stack([rand_fr(orc, 3, 3000 + k) for k in lengths])
    devs = [c.upload(p) for p in polys]
    for rep in range(2):
        tickets = c.open_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12), zeta, chal)
        c.submit()
        with pytest.raises(PlonkHipError):      # a second batch while tickets of the first are open is refused
            c.open_begin_batch(devs[:1], lengths[:1], blinds[:1].reshape(1, 12), zeta, chal)
        for i in (9, 0, 4, 1, 2, 3, 8, 7, 6, 5):
            w, winf, rv, ev = c.open_end(tickets[i])
            ow, oinf, orv = orc.kzg_open(g, gg, polys[i], blinds[i], zeta, chal)
            assert winf == oinf and np.array_equal(w, ow), lengths[i]
            want_ev = orc.poly_eval(polys[i], zeta) if lengths[i] else np.zeros(4, dtype=np.uint64)
            assert np.array_equal(rv, orv) and np.array_equal(ev, want_ev), lengths[i]
    for d in devs:
        c.free(d)
    # short and long block sizes in one batch, against the single-polynomial entry point (itself checked against the oracle at these
    # lengths by test_division_around_the_change_of_block_size) and, for the evaluations, against the oracle's Horner rule
    lengths = [(1 << 21) - 1, 1 << 21, (1 << 21) + 4095, 70001, 5]
    def residues(count, seed):                 # any value below 2^252 is a residue in Montgomery form
        a = np.random.default_rng(seed).integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    polys = [residues(k, 4000 + i) for i, k in enumerate(lengths)]
    blinds = np.stack([rand_fr(orc, 3, 5000 + i) for i in range(len(lengths))])
    devs = [c.upload(p) for p in polys]
    tickets = c.open_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12), zeta, chal)
    batch = [c.open_end(t) for t in tickets]
    for i, d in enumerate(devs):
        single = c.open_end(c.open_begin(d, lengths[i], blinds[i], zeta, chal))
        assert batch[i][1] == single[1] and all(np.array_equal(x, y) for x, y in zip((batch[i][0], batch[i][2], batch[i][3]), (single[0], single[2], single[3]))), lengths[i]
        assert np.array_equal(batch[i][3], orc.poly_eval(polys[i], zeta)), lengths[i]
        c.free(d)
    # Lengths between the narrow plan's limit and the limit for batched divisions: the members on the wide plan share their
    # launches in groups, beside the narrow group.  Checked through the trapdoor of this SRS (tau, gamma are known here):
    # W = ch * [(f(tau) - f(zeta)) + gamma (r(tau) - r(zeta))] / (tau - zeta) * G, evaluations by the oracle's Horner rule.
    lengths = [100000, 131072, 262144, 70001, 300000, 5, 65536]
    polys = [residues(k, 6000 + i) for i, k in enumerate(lengths)]
    blinds = np.stack([rand_fr(orc, 3, 6100 + i) for i in range(len(lengths))])
    devs = [c.upload(p) for p in polys]
    rmod = pm.R_MOD
    z_i, ch_i = orc.fr_to_ints(zeta)[0], orc.fr_to_ints(chal)[0]
    inv_d = pow((TAU - z_i) % rmod, rmod - 2, rmod)
    for rep in range(2):
        tickets = c.open_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12), zeta, chal)
        for i in (3, 0, 6, 2, 5, 1, 4):
            w, winf, rv, ev = c.open_end(tickets[i])
            f_tau, f_z = orc.fr_to_ints(orc.poly_eval(polys[i], tau))[0], orc.fr_to_ints(orc.poly_eval(polys[i], zeta))[0]
            b_tau, b_z = orc.fr_to_ints(orc.poly_eval(blinds[i], tau))[0], orc.fr_to_ints(orc.poly_eval(blinds[i], zeta))[0]
            k = ch_i * ((f_tau - f_z) + GSCAL * (b_tau - b_z)) % rmod * inv_d % rmod
            assert orc.point_to_ints(w, winf) == (pm.g1_mul(k, pm.G1_GEN) if k else None), lengths[i]
            assert orc.fr_to_ints(ev)[0] == f_z and orc.fr_to_ints(rv)[0] == ch_i * b_z % rmod, lengths[i]
    # the same polynomials as commitments begun with one call: C = (f(tau) + gamma r(tau)) G
    tickets = c.commit_begin_batch(devs, lengths, blinds.reshape(len(lengths), 12))
    pts, inf, _, _ = c.end_many(tickets)
    for i in range(len(lengths)):
        k = (orc.fr_to_ints(orc.poly_eval(polys[i], tau))[0] + GSCAL * orc.fr_to_ints(orc.poly_eval(blinds[i], tau))[0]) % rmod
        assert orc.point_to_ints(pts[i, :8], inf[i]) == pm.g1_mul(k, pm.G1_GEN), lengths[i]
    for d in devs:
        c.free(d)
    c.close()


def test_polynomial_overwritten_after_begin(ctx, orc, srs):
    """The front work of every second ticket runs on a second stream.  A compute call that overwrites the polynomial after the
    *_begin call that reads it (a transform in place, a copy, an upload) is ordered behind that read: the commitments and openings
    below must be those of the original polynomials, whichever stream read them."""
    srs_len = len(srs["g"])
    k = min(4096, srs_len - 1)
    polys = [rand_fr(orc, k, 800 + i) for i in range(6)]
    blinds = [rand_fr(orc, 3, 820 + i) for i in range(6)]
    zeta, chal = rand_fr(orc, 2, 77)
    chal = orc.fr_from_ints([orc.fr_to_ints(chal)[0] & ((1 << 128) - 1)])[0]
    junk = rand_fr(orc, k, 899)
    d_junk = ctx.upload(junk)
    devs = [ctx.upload(p) for p in polys]
    tickets = []
    for i, (d, b) in enumerate(zip(devs, blinds)):
        tickets.append(ctx.commit_begin(d, k, b) if i % 3 != 2 else ctx.open_begin(d, k, b, zeta, chal))
        # overwrite at once, three ways in turn
        if i % 3 == 0:
            ctx._ck(ctx.lib.plk_copy_dev(ctx.h, C.c_void_p(d), C.c_void_p(d_junk), C.c_uint64(k * 32), C.c_uint64(k * 32)))
        elif i % 3 == 1:
            ctx.upload_into(d, junk)
        else:
            ctx._ck(ctx.lib.plk_copy_dev(ctx.h, C.c_void_p(d), C.c_void_p(d_junk), C.c_uint64(0), C.c_uint64(k * 32)))   # zero fill
    for i, (t, p, b) in enumerate(zip(tickets, polys, blinds)):
        if i % 3 != 2:
            got, inf = ctx.commit_end(t)
            want = orc.kzg_commit(srs["g"], srs["gg"], p, b)
            assert inf == want[1] and np.array_equal(got, want[0]), i
        else:
            w, winf, rv, ev = ctx.open_end(t)
            ow, oinf, orv = orc.kzg_open(srs["g"], srs["gg"], p, b, zeta, chal)
            assert winf == oinf and np.array_equal(w, ow) and np.array_equal(ev, orc.poly_eval(p, zeta)), i
    for d in devs + [d_junk]:
        ctx.free(d)


def test_kzg_open_lengths_around_the_division_blocks(orc):
    """The division by (X - zeta) works in blocks of 2048 coefficients (8192 from 2^21 coefficients on, see the scale tests) with a
    carry across blocks: lengths on both sides of one, two and several block boundaries, against the oracle's long division and MSM."""
    from plonk_amd.hip import Context
    c = Context(0)
    n = 1 << 16
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    c.srs_generate(tau, gs, n + 1)
    g = c.srs_read(0, 0, n + 1); gg = c.srs_read(1, 0, 3)
    for length in (2047, 2048, 2049, 4096, 4097, 6145, 8191, 8192, 8193, 16384, 16385, 24577, 40000, 65536, 65537):
        coeffs = rand_fr(orc, length, 90 + length)
        blind = rand_fr(orc, 3, 91 + length)
        zeta, chal = rand_fr(orc, 2, 92 + length)
        chal = orc.fr_from_ints([orc.fr_to_ints(chal)[0] & ((1 << 128) - 1)])[0]
        w, winf, rv, ev = c.kzg_open(coeffs, blind, zeta, chal)
        ow, oinf, orv = orc.kzg_open(g, gg, coeffs, blind, zeta, chal)
        assert winf == oinf and np.array_equal(w, ow), length
        assert np.array_equal(rv, orv) and np.array_equal(ev, orc.poly_eval(coeffs, zeta)), length
    # zeta = 0 and zeta = 1: the quotient is a shift / a running sum of the coefficients
    coeffs = rand_fr(orc, 20000, 5); blind = rand_fr(orc, 3, 6)
    for z in (0, 1, pm.R_MOD - 1):
        zeta = orc.fr_from_ints([z])[0]; chal = orc.fr_from_ints([0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF])[0]
        w, winf, rv, ev = c.kzg_open(coeffs, blind, zeta, chal)
        ow, oinf, orv = orc.kzg_open(g, gg, coeffs, blind, zeta, chal)
        assert winf == oinf and np.array_equal(w, ow) and np.array_equal(rv, orv) and np.array_equal(ev, orc.poly_eval(coeffs, zeta)), z
    c.close()


def test_failure_behaviour_follows_the_reference(orc):
    """Where the reference panics the C ABI returns a status (and the context stays usable):
    TooManyCoefficients (PCS::commit on a polynomial longer than the key, src/prover.rs:150), index out of bounds for a sigma entry
    >= 3n (src/circuit.rs:508-510), .inverse().unwrap() on a zero denominator (src/circuit.rs:523), the sigma-length assertion of
    interpolate_permutation_polynomials (src/poly_utils.rs:22); plus misuse of tickets, which the reference has no counterpart for."""
    from plonk_amd.hip import Context, PlonkHipError
    c = Context(0)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    # no SRS yet
    with pytest.raises(PlonkHipError) as e:
        c.msm(rand_fr(orc, 4, 1))
    assert e.value.status == "PLK_ERR_INVALID_ARG"
    c.srs_generate(tau, gs, 65)
    blind = rand_fr(orc, 3, 2)
    with pytest.raises(PlonkHipError) as e:
        c.kzg_commit(rand_fr(orc, 66, 3), blind)
    assert e.value.status == "PLK_ERR_SRS_TOO_SHORT"
    with pytest.raises(PlonkHipError) as e:
        c.msm(rand_fr(orc, 10, 4), base_offset=60)
    assert e.value.status == "PLK_ERR_SRS_TOO_SHORT"
    # the context still works after a refused call
    coeffs = rand_fr(orc, 65, 5)
    g = c.srs_read(0, 0, 65); gg = c.srs_read(1, 0, 3)
    got, inf = c.kzg_commit(coeffs, blind)
    want, winf = orc.kzg_commit(g, gg, coeffs, blind)
    assert inf == winf and np.array_equal(got, want)

    # grand product: sigma entry out of range, then a zero factor in the denominator
    v, gates = pm.synthetic_circuit(3, seed=5)
    inp = pm.prepare_inputs(v, gates)
    w = orc.fr_from_ints(inp["witness_flat"]); sigma = np.array(inp["sigma"], dtype=np.uint64)
    beta, gamma = orc.fr_from_ints([11, 13])
    bad = sigma.copy(); bad[7] = 3 * inp["n"]
    with pytest.raises(PlonkHipError) as e:
        c.grand_product(w, bad, beta, gamma)
    assert e.value.status == "PLK_ERR_INDEX_OUT_OF_RANGE"
    # gamma := -(w_j + beta j) makes the factor of position j vanish: the reference's inverse().unwrap() panics
    j = int(sigma[4])
    zero_gamma = orc.fr_from_ints([(-(inp["witness_flat"][j] + 11 * j)) % pm.R_MOD])[0]
    with pytest.raises(PlonkHipError) as e:
        c.grand_product(w, sigma, beta, zero_gamma)
    assert e.value.status == "PLK_ERR_INVERSE_OF_ZERO"
    zc = c.grand_product(w, sigma, beta, gamma)
    assert np.array_equal(zc[1] if isinstance(zc, tuple) else zc, orc.grand_product(w, sigma, beta, gamma)[1])

    # permutation interpolation: more sigma entries than the domain holds; an entry outside the domain
    with pytest.raises(PlonkHipError) as e:
        c.interpolate_permutation(np.arange(33, dtype=np.uint64), 5)
    assert e.value.status == "PLK_ERR_INVALID_ARG"
    # an entry beyond the domain is not an error: domain.element(i) is w^i for any i (src/poly_utils.rs:25), so 32 + 5 acts as 5
    wrapped = np.arange(24, dtype=np.uint64); wrapped[3] = 32 + 5
    inside = np.arange(24, dtype=np.uint64); inside[3] = 5
    got_id, got_sigma = c.interpolate_permutation(wrapped, 5)
    want_id, want_sigma = c.interpolate_permutation(inside, 5)
    assert np.array_equal(got_id, want_id) and np.array_equal(got_sigma, want_sigma)
    x_poly = np.zeros((32, 4), np.uint64); x_poly[1] = orc.fr_from_ints([1])[0]
    dom = orc.ntt(x_poly)
    assert np.array_equal(got_sigma, orc.ntt(dom[np.concatenate([inside, np.arange(24, 32, dtype=np.uint64)]).astype(np.int64)], inverse=True))

    # tickets: ending one twice, ending one that was never handed out, more tickets than slots
    d = c.upload(coeffs)
    t = c.commit_begin(d, 65, blind)
    out, inf = c.commit_end(t)
    assert np.array_equal(out, want)
    with pytest.raises(PlonkHipError) as e:
        c.commit_end(t)
    assert e.value.status == "PLK_ERR_INVALID_ARG"
    with pytest.raises(PlonkHipError) as e:
        c.commit_end(99)
    assert e.value.status == "PLK_ERR_INVALID_ARG"
    tickets = []
    with pytest.raises(PlonkHipError) as e:
        for _ in range(64):
            tickets.append(c.commit_begin(d, 65, blind))
    assert e.value.status == "PLK_ERR_INVALID_ARG" and "slots are busy" in str(e.value) and len(tickets) >= 9
    for t in tickets:
        out, inf = c.commit_end(t)
        assert np.array_equal(out, want)
    # a zero coset generator has no inverse
    with pytest.raises(PlonkHipError) as e:
        c.ntt(rand_fr(orc, 16, 6), inverse=True, coset=orc.fr_from_ints([0])[0])
    assert e.value.status == "PLK_ERR_INVERSE_OF_ZERO"
    c.free(d)
    c.close()


def test_many_coset_generators(ctx, orc):
    """Large coset transforms keep a table of per-element factors per generator; a caller that uses many generators gets correct
    results throughout and the number of tables stays bounded (twenty generators here, eight tables kept)."""
    log_n = 12
    a = rand_fr(orc, 1 << log_n, 123)
    d = ctx.upload(a)
    for k in range(20):
        g = orc.fr_from_ints([5 + 2 * k])[0]
        ctx.ntt_dev(d, log_n, coset=g)
        if k % 5 == 0:
            assert np.array_equal(ctx.download(d, (1 << log_n, 4)), orc.ntt(a, coset=g)), k
        ctx.ntt_dev(d, log_n, inverse=True, coset=g)
    assert np.array_equal(ctx.download(d, (1 << log_n, 4)), a)
    ctx.free(d)


def test_upload_enqueue_from_ordinary_and_page_locked_memory(ctx):
    """plk_upload_enqueue: ordinary arrays travel through the library's staging ring (helper threads, 16 MB buffers, four of them),
    page-locked ones directly; sizes below, at and across the buffer size, several uploads in a row, odd byte counts."""
    rng = np.random.default_rng(77)
    sizes = [5, 1 << 20, (1 << 20) + 3, (16 << 20) - 1, 16 << 20, (16 << 20) + 4097, (80 << 20) + 11]
    dev = [ctx.malloc(n) for n in sizes]
    host = [rng.integers(0, 256, size=n, dtype=np.uint8) for n in sizes]
    for d, h in zip(dev, host):
        ctx.upload_enqueue(d, h)
    ctx.sync()
    for d, h in zip(dev, host):
        assert np.array_equal(ctx.download(d, h.shape, np.uint8), h)
    # the same buffers again from page-locked memory, with other contents
    pinned = []
    for d, h in zip(dev, host):
        p = ctx.host_empty(h.shape, np.uint8); p[...] = h[::-1]; pinned.append(p)
        ctx.upload_enqueue(d, p)
    ctx.sync()
    for d, p in zip(dev, pinned):
        assert np.array_equal(ctx.download(d, p.shape, np.uint8), p)
    for p in pinned:
        ctx.host_free(p)
    for d in dev:
        ctx.free(d)
