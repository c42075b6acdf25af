"""CPU restatement of the randomness the reference's driver consumes (TEST INFRASTRUCTURE - never imported by plonk_amd).

Pure Python integers; everything here is small (config 1 of the reference: 33 powers, one Poseidon permutation).

What is restated, and from where:
  * ark_std::test_rng() (ark-std 0.5 `test_rng_helper`, the deterministic branch that `DETERMINISTIC_TEST_RNG=1` selects):
    rand 0.8 `StdRng` = rand_chacha 0.3 `ChaCha12Rng` seeded with [1,0,0,0, 23,0,0,0, 200,1,0,0, 210,30,0,0, 0 x 16].
    Used by /root/reference/src/main.rs:25 and /root/reference/src/prover.rs:48.
  * ark-ff 0.5 `UniformRand for Fp`: four u64 limbs, top two bits cleared, rejected when >= modulus, taken AS the
    Montgomery representation (so the value is limbs * 2^-256).  /root/reference/src/main.rs:136-137,174-176, prover.rs:53.
  * ark-ec 0.5 `Distribution<Projective> for Standard` (short Weierstrass): x <- BaseField::rand, greatest <- bool,
    point from x (the larger y when `greatest`), times the cofactor.  Reached from KZG10::setup, main.rs:182.
  * ark-poly-commit 0.5 `KZG10::setup` + `MarlinKZG10::trim(pp, d, 1, None)`: beta, g, gamma_g, h drawn in that order;
    powers_of_g[i] = beta^i g (i <= d), powers_of_gamma_g[j] = beta^j gamma_g (j < 3), vk = (g, gamma_g, h, beta h).  main.rs:182-183.
  * ark-crypto-primitives 0.5 `PoseidonSponge` as configured by `test_sponge` (/root/reference/src/prover.rs:36-59) and the
    one 128-bit challenge `MarlinKZG10::open` squeezes from it (prover.rs:285-303).

Pins: the ChaCha core against RFC 8439 2.3.2 (20 rounds) and the generator against rand 0.8's own value-stability test
(`StdRng::from_seed(<the seed above>).next_u64() == 10719222850664546238`).  The rest follows the published algorithms of
the crates named above, which are not available in this environment: parity unpinned.
"""
from __future__ import annotations

import struct

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MONT_INV_R = pow(1 << 256, -1, R_MOD)
MONT_INV_Q = pow(1 << 256, -1, Q_MOD)
G2_COFACTOR = 2 * Q_MOD - R_MOD            # #E'(Fq2) = r (2q - r) for a BN curve
TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)


# ------------------------------------------------------------------------------------------------ ChaCha / StdRng
def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF


def _quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key: bytes, counter: int, stream: int, rounds: int) -> list:
    """One 16-word block; words 12,13 = 64-bit block counter, 14,15 = 64-bit stream id (rand_chacha's layout)."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key)) + [
        counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, stream & 0xFFFFFFFF, (stream >> 32) & 0xFFFFFFFF]
    w = init[:]
    for _ in range(rounds // 2):
        _quarter(w, 0, 4, 8, 12); _quarter(w, 1, 5, 9, 13); _quarter(w, 2, 6, 10, 14); _quarter(w, 3, 7, 11, 15)
        _quarter(w, 0, 5, 10, 15); _quarter(w, 1, 6, 11, 12); _quarter(w, 2, 7, 8, 13); _quarter(w, 3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(w, init)]


class StdRng:
    """rand 0.8 StdRng: ChaCha12, a buffer of four blocks (64 words) read through rand_core's BlockRng."""

    def __init__(self, seed: bytes = TEST_RNG_SEED):
        assert len(seed) == 32
        self.key = seed
        self.counter = 0
        self.buf = [0] * 64
        self.index = 64           # empty

    def _refill(self, index):
        self.buf = []
        for _ in range(4):
            self.buf += chacha_block(self.key, self.counter, 0, 12)
            self.counter += 1
        self.index = index

    def next_u32(self) -> int:
        if self.index >= 64:
            self._refill(0)
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        i = self.index
        if i < 63:
            self.index += 2
            return self.buf[i] | (self.buf[i + 1] << 32)
        if i >= 64:
            self._refill(2)
            return self.buf[0] | (self.buf[1] << 32)
        lo = self.buf[63]
        self._refill(1)
        return lo | (self.buf[0] << 32)

    def gen_bool(self) -> bool:
        """`rng.gen::<bool>()`: the sign bit of one u32"""
        return (self.next_u32() >> 31) == 1


def test_rng() -> StdRng:
    return StdRng(TEST_RNG_SEED)


# ------------------------------------------------------------------------------------------------ field and curve sampling
def _fp_rand(rng: StdRng, modulus: int, mont_inv: int) -> int:
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 2
        raw = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        if raw < modulus:
            return raw * mont_inv % modulus


def fr_rand(rng: StdRng) -> int:
    return _fp_rand(rng, R_MOD, MONT_INV_R)


def fq_rand(rng: StdRng) -> int:
    return _fp_rand(rng, Q_MOD, MONT_INV_Q)


def fq_sqrt(a: int):
    r = pow(a, (Q_MOD + 1) // 4, Q_MOD)
    return r if r * r % Q_MOD == a % Q_MOD else None


def g1_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q_MOD == 0: return None
        l = 3 * p[0] * p[0] * pow(2 * p[1], -1, Q_MOD) % Q_MOD
    else:
        l = (q[1] - p[1]) * pow(q[0] - p[0], -1, Q_MOD) % Q_MOD
    x = (l * l - p[0] - q[0]) % Q_MOD
    return (x, (l * (p[0] - x) - p[1]) % Q_MOD)


def g1_mul(p, k: int):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1": acc = g1_add(acc, p)
    return acc


def g1_rand(rng: StdRng):
    while True:
        x = fq_rand(rng)
        greatest = rng.gen_bool()
        y = fq_sqrt((x * x * x + 3) % Q_MOD)
        if y is None:
            continue
        small, large = sorted((y, (-y) % Q_MOD))
        return (x, large if greatest else small)      # cofactor 1


# Fq2 = Fq[i]/(i^2 + 1), elements (c0, c1)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def f2_add(a, b): return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)
def f2_sub(a, b): return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)
def f2_neg(a): return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q_MOD)
    return (a[0] * d % Q_MOD, (-a[1]) * d % Q_MOD)


def _legendre_is_qr(a):      # ark's `legendre().is_qr()`: 0 is not a residue
    return a % Q_MOD != 0 and pow(a, (Q_MOD - 1) // 2, Q_MOD) == 1


def f2_sqrt(a):
    """ark-ff QuadExtField::sqrt (complex method); which of the two roots comes out does not matter to its callers here"""
    if a[1] == 0:
        if a[0] == 0: return (0, 0)
        if _legendre_is_qr(a[0]): return (fq_sqrt(a[0]), 0)
        r = fq_sqrt(a[0] * pow(Q_MOD - 1, -1, Q_MOD) % Q_MOD)       # c0 / nonresidue, nonresidue = -1
        return None if r is None else (0, r)
    alpha = fq_sqrt((a[0] * a[0] + a[1] * a[1]) % Q_MOD)            # sqrt of the norm
    if alpha is None: return None
    two_inv = (Q_MOD + 1) // 2
    delta = (alpha + a[0]) * two_inv % Q_MOD
    if not _legendre_is_qr(delta) and delta != 0:
        delta = (delta - alpha) % Q_MOD
    c0 = fq_sqrt(delta)
    if c0 is None or c0 == 0: return None
    cand = (c0, a[1] * two_inv % Q_MOD * pow(c0, -1, Q_MOD) % Q_MOD)
    return cand if f2_mul(cand, cand) == (a[0] % Q_MOD, a[1] % Q_MOD) else None


G2_B = f2_mul((3, 0), f2_inv((9, 1)))          # twist coefficient 3 / (9 + i)


def g2_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if f2_add(p[1], q[1]) == (0, 0): return None
        xx = f2_mul(p[0], p[0])
        l = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(p[1], p[1])))
    else:
        l = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x = f2_sub(f2_sub(f2_mul(l, l), p[0]), q[0])
    return (x, f2_sub(f2_mul(l, f2_sub(p[0], x)), p[1]))


def g2_mul(p, k: int):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1": acc = g2_add(acc, p)
    return acc


def g2_rand(rng: StdRng):
    while True:
        x = (fq_rand(rng), fq_rand(rng))
        greatest = rng.gen_bool()
        y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), G2_B))
        if y is None:
            continue
        ny = f2_neg(y)
        # ark-ff orders Fq2 by c1, then c0
        small, large = sorted((y, ny), key=lambda v: (v[1], v[0]))
        return g2_mul((x, large if greatest else small), G2_COFACTOR)


# ------------------------------------------------------------------------------------------------ KZG10::setup + trim
def kzg_setup(max_degree: int, rng: StdRng) -> dict:
    beta = fr_rand(rng)
    g = g1_rand(rng)
    gamma_g = g1_rand(rng)
    h = g2_rand(rng)
    powers, gamma_powers = [], []
    cur = 1
    for i in range(max_degree + 1):
        powers.append(g1_mul(g, cur))
        if i < 3: gamma_powers.append(g1_mul(gamma_g, cur))
        cur = cur * beta % R_MOD
    return {"beta": beta, "g": g, "gamma_g": gamma_g, "h": h, "beta_h": g2_mul(h, beta), "powers_of_g": powers, "powers_of_gamma_g": gamma_powers}


# ------------------------------------------------------------------------------------------------ Poseidon sponge of test_sponge
FULL_ROUNDS, PARTIAL_ROUNDS, ALPHA = 8, 31, 17
MDS = [[1, 0, 1], [1, 1, 0], [0, 1, 1]]


def poseidon_round_constants() -> list:
    rng = test_rng()
    return [[fr_rand(rng) for _ in range(3)] for _ in range(FULL_ROUNDS + PARTIAL_ROUNDS)]


def poseidon_permute(state: list, ark: list) -> list:
    return poseidon_permute_rounds(state, ark, FULL_ROUNDS + PARTIAL_ROUNDS)


def poseidon_permute_rounds(state: list, ark: list, rounds: int) -> list:
    """the first `rounds` rounds of the permutation: 4 full, 31 partial (S-box on state[0] only), 4 full; each round adds its
    constants, applies x^17 and multiplies by the MDS matrix"""
    s = list(state)
    half = FULL_ROUNDS // 2
    for rnd in range(rounds):
        s = [(x + c) % R_MOD for x, c in zip(s, ark[rnd])]
        if rnd < half or rnd >= half + PARTIAL_ROUNDS:
            s = [pow(x, ALPHA, R_MOD) for x in s]
        else:
            s[0] = pow(s[0], ALPHA, R_MOD)
        s = [sum(m * x for m, x in zip(row, s)) % R_MOD for row in MDS]
    return s


def opening_challenge() -> int:
    """What `PCS::open` draws: squeeze_field_elements_with_sizes([Truncated(128)]) on a fresh sponge = one permutation of the
    zero state, the first rate element (state[capacity + 0] = state[1]), its low 128 bits."""
    state = poseidon_permute([0, 0, 0], poseidon_round_constants())
    return state[1] & ((1 << 128) - 1)


# ------------------------------------------------------------------------------------------------ the draws of main()
def main_draws(n_rows: int = 4, max_degree: int = 32) -> dict:
    """The rng of /root/reference/src/main.rs from line 25 to the proof: 2 placeholders per wire (main.rs:136-137), three more
    (:174-176), PCS::setup (:182), then 27 blinding coefficients inside the prover (three per PCS::commit, prover.rs:150-156,210,253)."""
    rng = test_rng()
    discarded = [fr_rand(rng) for _ in range(2 * 3 * n_rows + 3)]
    srs = kzg_setup(max_degree, rng)
    blinds = [fr_rand(rng) for _ in range(27)]
    return {"discarded": discarded, "srs": srs, "blinds": blinds, "challenge": opening_challenge()}
