"""Big-integer model of the reference prover (TEST INFRASTRUCTURE - never imported by the product).

Pure-Python restatement, with Python ints, of what the reference computes on its prover
hot path.  It exists to (a) generate the small golden vectors under tests/golden/ and
(b) pin the faster C++ restatement in oracle/oracle.cpp, which in turn checks the HIP
library.  PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures and
cannot be built in this environment (no Rust toolchain, arithmetic lives in un-vendored
arkworks 0.5 / merlin 3.0 crates), so the only external anchors are the published merlin
known-answer test, the BN254 constants, and the RNG-independent values of the reference's
example circuit.

Reference call sites restated here
  src/prover.rs:119-336     create_plonk_proof_with_transcript (round flow, absorb order)
  src/circuit.rs:80-112     PermutationLayout::compute_sigma_mapping
  src/circuit.rs:216-259    CircuitBuilder::generate_witness_table
  src/circuit.rs:343-405    Circuit::build_permutation_constraint
  src/circuit.rs:442-489    Circuit::build_quotient_polynomial
  src/circuit.rs:491-544    Circuit::build_grand_product
  src/poly_utils.rs:7-31    interpolate_selector / interpolate_permutation_polynomials
  src/transcript.rs:7-52    absorb helpers, challenge_scalar (64 bytes LE mod r)
  src/main.rs:29-100,187-191  example circuit and how the prover inputs are prepared
Third-party semantics (ark-poly-commit 0.5 kzg10/marlin_pc, ark-poly 0.5, ark-serialize 0.5,
merlin 3.0) are restated from their published algorithms.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------- constants
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq
TWO_ADICITY = 28
FR_GENERATOR = 5
ROOT_2_28 = pow(FR_GENERATOR, (R_MOD - 1) >> TWO_ADICITY, R_MOD)
G1_GEN = (1, 2)
MONT_R = 1 << 256


def root_of_unity(n: int) -> int:
    """ark-poly Radix2EvaluationDomain group generator for size n (power of two)."""
    assert n & (n - 1) == 0 and n >= 1
    k = n.bit_length() - 1
    assert k <= TWO_ADICITY
    w = ROOT_2_28
    for _ in range(TWO_ADICITY - k):
        w = w * w % R_MOD
    return w


def finv(x: int, p: int = R_MOD) -> int:
    return pow(x, -1, p)


# ----------------------------------------------------------------------------- limbs / ABI
def to_mont_limbs(x: int, p: int = R_MOD):
    """4 little-endian u64 limbs of x*2^256 mod p (arkworks' in-memory Fp payload)."""
    v = x * MONT_R % p
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_mont_limbs(l, p: int = R_MOD) -> int:
    v = sum(int(l[i]) << (64 * i) for i in range(4))
    return v * finv(MONT_R % p, p) % p


# ----------------------------------------------------------------------------- G1 (affine, None = infinity)
def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q_MOD)


def g1_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % Q_MOD == 0:
            return None
        lam = 3 * x1 * x1 * finv(2 * y1, Q_MOD) % Q_MOD
    else:
        lam = (y2 - y1) * finv(x2 - x1, Q_MOD) % Q_MOD
    x3 = (lam * lam - x1 - x2) % Q_MOD
    y3 = (lam * (x1 - x3) - y1) % Q_MOD
    return (x3, y3)


def g1_mul(k: int, P):
    k %= R_MOD
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, P)
        P = g1_add(P, P)
        k >>= 1
    return acc


def g1_on_curve(P) -> bool:
    return P is None or (P[1] * P[1] - P[0] ** 3 - 3) % Q_MOD == 0


def msm(scalars, points):
    """Naive sum of k_i * P_i."""
    acc = None
    for k, P in zip(scalars, points):
        acc = g1_add(acc, g1_mul(k, P))
    return acc


# ----------------------------------------------------------------------------- polynomials / domains
def trim(c):
    """DensePolynomial::from_coefficients_vec: strip trailing zeros."""
    c = [x % R_MOD for x in c]
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_eval(c, x):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % R_MOD
    return acc


def ntt(vals, n=None, inverse=False, coset=None):
    """Radix-2 domain transform, natural order in and out (ark-poly fft / ifft semantics).

    Shorter inputs are zero padded to n.  inverse multiplies by n^-1.  coset=g evaluates
    at g*w^i (forward) or interpolates from those points (inverse)."""
    n = n or len(vals)
    a = [v % R_MOD for v in vals] + [0] * (n - len(vals))
    assert len(a) == n and n & (n - 1) == 0
    w = root_of_unity(n)
    if inverse:
        w = finv(w)
    if coset is not None and not inverse:
        gp = 1
        for i in range(n):
            a[i] = a[i] * gp % R_MOD
            gp = gp * coset % R_MOD
    out = _fft_rec(a, w)
    if inverse:
        ninv = finv(n)
        out = [v * ninv % R_MOD for v in out]
        if coset is not None:
            gi = finv(coset)
            gp = 1
            for i in range(n):
                out[i] = out[i] * gp % R_MOD
                gp = gp * gi % R_MOD
    return out


def _fft_rec(a, w):
    n = len(a)
    if n == 1:
        return a
    ev = _fft_rec(a[0::2], w * w % R_MOD)
    od = _fft_rec(a[1::2], w * w % R_MOD)
    out = [0] * n
    t = 1
    for i in range(n // 2):
        x = od[i] * t % R_MOD
        out[i] = (ev[i] + x) % R_MOD
        out[i + n // 2] = (ev[i] - x) % R_MOD
        t = t * w % R_MOD
    return out


def dft_naive(vals, inverse=False):
    n = len(vals)
    w = root_of_unity(n)
    if inverse:
        w = finv(w)
    out = [sum(vals[j] * pow(w, i * j, R_MOD) for j in range(n)) % R_MOD for i in range(n)]
    if inverse:
        ninv = finv(n)
        out = [v * ninv % R_MOD for v in out]
    return out


def div_linear(c, z):
    """Quotient of c(X) / (X - z), remainder dropped (ark-poly DensePolynomial division)."""
    if len(c) < 2:
        return []
    q = [0] * (len(c) - 1)
    carry = 0
    for i in range(len(c) - 1, 0, -1):
        carry = (c[i] + carry * z) % R_MOD
        q[i - 1] = carry
    return trim(q)


def divide_by_vanishing(c, n):
    """(quotient, remainder) of c(X) / (X^n - 1) (ark-poly divide_by_vanishing_poly)."""
    c = trim(c)
    if len(c) < n:
        return [], c
    q = list(c[n:])
    for i in range(1, len(c) // n):
        for k, v in enumerate(c[n * (i + 1):]):
            q[k] = (q[k] + v) % R_MOD
    r = list(c[:n])
    for k, v in enumerate(q[:n]):
        r[k] = (r[k] + v) % R_MOD
    return trim(q), trim(r)


# ----------------------------------------------------------------------------- serialization (ark-serialize 0.5, compressed)
def fr_bytes(x: int) -> bytes:
    return (x % R_MOD).to_bytes(32, "little")


def g1_compressed(P) -> bytes:
    """32-byte LE x with flags in the top byte: 0x80 if y > -y, 0x40 for infinity."""
    if P is None:
        b = bytearray(32)
        b[31] |= 0x40
        return bytes(b)
    x, y = P
    b = bytearray(x.to_bytes(32, "little"))
    if y > (Q_MOD - y) % Q_MOD:
        b[31] |= 0x80
    return bytes(b)


def commitment_bytes(P) -> bytes:
    """marlin_pc::Commitment{comm, shifted_comm: None}: 32 + 1 bytes."""
    return g1_compressed(P) + b"\x00"


def opening_bytes(W, random_v) -> bytes:
    """kzg10::Proof{w, random_v: Some(v)}: 32 + 1 + 32 bytes."""
    return g1_compressed(W) + b"\x01" + fr_bytes(random_v)


# ----------------------------------------------------------------------------- merlin 3.0 (STROBE-128 / Keccak-f[1600])
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(state: bytearray) -> None:
    A = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = A[x][y].to_bytes(8, "little")


class Strobe128:
    R = 166
    I, A, C, T, M, K = 1, 2, 4, 8, 16, 32

    def __init__(self, label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.st, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(label, False)

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        keccak_f1600(self.st)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data: bytes):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray(n)
        for i in range(n):
            out[i] = self.st[self.pos]
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert flags == self.cur_flags
            return
        assert flags & self.T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (self.C | self.K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(self.M | self.A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(self.A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool) -> bytes:
        self._begin_op(self.I | self.A | self.C, more)
        return self._squeeze(n)


class Transcript:
    """merlin::Transcript."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, msg: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(msg).to_bytes(4, "little"), True)
        self.strobe.ad(msg, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)

    def challenge_scalar(self, label: bytes) -> int:
        """src/transcript.rs:14-18: 64 squeezed bytes, little-endian, reduced mod r."""
        return int.from_bytes(self.challenge_bytes(label, 64), "little") % R_MOD


# ----------------------------------------------------------------------------- deterministic inputs (this build's own generator)
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def next_fr(self) -> int:
        """4 limbs, top two bits cleared, rejected while >= r (canonical value)."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next_u64() << (64 * i)
            v &= (1 << 254) - 1
            if v < R_MOD:
                return v


def make_srs(num_g: int, tau: int, gscal: int):
    """powers_of_g[i] = tau^i * G (i < num_g); powers_of_gamma_g[j] = gscal * tau^j * G (j < 3)."""
    g = []
    t = 1
    for _ in range(num_g):
        g.append(g1_mul(t, G1_GEN))
        t = t * tau % R_MOD
    gg = [g1_mul(gscal * pow(tau, j, R_MOD) % R_MOD, G1_GEN) for j in range(3)]
    return g, gg


# ----------------------------------------------------------------------------- circuits
ADD, MUL = 0, 1


def example_circuit():
    """src/main.rs:29-39: the 4-gate chain.  Returns (variables, gates[(type, a, b, out)])."""
    variables = [3, 4]
    gates = []

    def add_gate(t, a, b):
        v = (variables[a] + variables[b]) % R_MOD if t == ADD else variables[a] * variables[b] % R_MOD
        variables.append(v)
        gates.append((t, a, b, len(variables) - 1))
        return len(variables) - 1

    a, b = 0, 1
    c = add_gate(ADD, a, b)
    variables.append(2)
    d = len(variables) - 1
    e = add_gate(MUL, c, d)
    variables.append(5)
    f = len(variables) - 1
    g = add_gate(ADD, e, f)
    add_gate(MUL, g, a)
    return variables, gates


def synthetic_circuit(log_n: int, seed: int = 1):
    """Random arithmetic circuit with exactly 2^log_n gates.

    v0, v1 uniform in Fr; gate 0 uses (v0, v1); gate i>0 takes the previous gate's output
    as its left input and a uniformly chosen earlier variable as its right input; the type
    is Add or Mul with probability 1/2.  The last row is re-drawn until none of its three
    values is zero (the reference indexes out of bounds otherwise: src/prover.rs:229-231)."""
    n = 1 << log_n
    rng = SplitMix64(seed)
    variables = [rng.next_fr(), rng.next_fr()]
    gates = []
    for i in range(n):
        while True:
            t = rng.next_u64() & 1
            if i == 0:
                a, b = 0, 1
            else:
                a = len(variables) - 1
                b = rng.next_u64() % (len(variables) - 1)
            v = (variables[a] + variables[b]) % R_MOD if t == ADD else variables[a] * variables[b] % R_MOD
            if i == n - 1 and (variables[a] == 0 or variables[b] == 0 or v == 0):
                continue
            break
        variables.append(v)
        gates.append((t, a, b, len(variables) - 1))
    return variables, gates


def witness_table(variables, gates, n):
    """src/circuit.rs:216-259 (columns zero padded to the domain size)."""
    a = [variables[g[1]] for g in gates] + [0] * (n - len(gates))
    b = [variables[g[2]] for g in gates] + [0] * (n - len(gates))
    c = [variables[g[3]] for g in gates] + [0] * (n - len(gates))
    qa = [1 if g[0] == ADD else 0 for g in gates] + [0] * (n - len(gates))
    qm = [1 if g[0] == MUL else 0 for g in gates] + [0] * (n - len(gates))
    return a, b, c, qa, qm


def sigma_mapping(gates, n):
    """src/circuit.rs:262-276 + 80-112: flat index = 3*row + col; each variable's uses form one cycle."""
    uses = {}
    for row, (_, a, b, c) in enumerate(gates):
        uses.setdefault(a, []).append(3 * row)
        uses.setdefault(b, []).append(3 * row + 1)
        uses.setdefault(c, []).append(3 * row + 2)
    sigma = list(range(3 * n))
    for _, u in uses.items():
        for i in range(len(u)):
            sigma[u[i]] = u[(i + 1) % len(u)]
    return sigma


def interpolate_selector(values):
    """src/poly_utils.rs:7-11."""
    n = 1
    while n < len(values):
        n *= 2
    return trim(ntt(values, n, inverse=True))


def interpolate_permutation_polynomials(sigma, n4):
    """src/poly_utils.rs:14-31 on a domain of size n4."""
    sp = list(sigma) + list(range(len(sigma), n4))
    assert len(sp) == n4
    w = root_of_unity(n4)
    pw = [1] * n4
    for i in range(1, n4):
        pw[i] = pw[i - 1] * w % R_MOD
    s_id = trim(ntt(pw, n4, inverse=True))
    s_sigma = trim(ntt([pw[i] for i in sp], n4, inverse=True))
    return s_id, s_sigma


def prepare_inputs(variables, gates):
    """src/main.rs:42-100,187-191: everything create_plonk_proof_with_transcript is called with."""
    n = 1
    while n < len(gates):
        n *= 2
    n4 = 1
    while n4 < 3 * n:
        n4 *= 2
    a_col, b_col, c_col, qa, qm = witness_table(variables, gates, n)
    sigma = sigma_mapping(gates, n)
    s_id, s_sigma = interpolate_permutation_polynomials(sigma, n4)
    witness_flat = []
    for i in range(n):
        witness_flat += [a_col[i], b_col[i], c_col[i]]
    return dict(
        n=n, n4=n4,
        a=trim(a_col), b=trim(b_col), c=trim(c_col),          # values used AS coefficients (main.rs:187-189)
        q_add=trim(ntt(qa, n, inverse=True)), q_mul=trim(ntt(qm, n, inverse=True)),
        s_id=s_id, s_sigma=s_sigma, witness_flat=witness_flat, sigma=sigma,
        cols=(a_col, b_col, c_col, qa, qm),
    )


# ----------------------------------------------------------------------------- round builders
def build_grand_product(witness_flat, sigma, n, beta, gamma):
    """src/circuit.rs:491-544 with s_id_vals[j] = j (src/prover.rs:205).  Returns (z values, z coeffs)."""
    z = [1] * (n + 1)
    for i in range(n):
        num = den = 1
        for k in range(3):
            j = 3 * i + k
            num = num * (witness_flat[j] + beta * j + gamma) % R_MOD
            s = sigma[j]
            den = den * (witness_flat[s] + beta * s + gamma) % R_MOD
        z[i + 1] = z[i] * num % R_MOD * finv(den) % R_MOD
    return z[:n], trim(ntt(z[:n], n, inverse=True))


def gate_values(a_col, b_col, c_col, qa_vals, qm_vals):
    """src/circuit.rs:446-453."""
    return [(qa_vals[i] * (a_col[i] + b_col[i] - c_col[i]) + qm_vals[i] * (a_col[i] * b_col[i] - c_col[i])) % R_MOD
            for i in range(len(a_col))]


def permutation_constraint_values(a_col, b_col, c_col, z_vals, sigma, n, beta, gamma):
    """src/circuit.rs:343-379."""
    out = [0] * n
    for i in range(n):
        w = (a_col[i], b_col[i], c_col[i])
        lhs = rhs = 1
        for k in range(3):
            lhs = lhs * (w[k] + beta * (3 * i + k) + gamma) % R_MOD
            rhs = rhs * (w[k] + beta * sigma[3 * i + k] + gamma) % R_MOD
        if i < n - 1:
            out[i] = (z_vals[i] * lhs - z_vals[i + 1] * rhs) % R_MOD
        else:
            out[i] = (z_vals[i] * lhs % R_MOD * finv(rhs) - 1) % R_MOD
    return out


def public_input_values(a_col, n, alpha, pi_rows, pi_values):
    """src/circuit.rs:419-436: constraint[row] = alpha * (a_vals[row] - pi_value), assigned in the order of the public inputs (a later
    one on the same row replaces an earlier one); pi_rows[k] is the first gate row that uses public variable k (:424-429)."""
    out = [0] * n
    for row, v in zip(pi_rows, pi_values):
        out[row] = alpha * (a_col[row] - v) % R_MOD
    return out


class NotDivisible(Exception):
    pass


def build_quotient(a_col, b_col, c_col, qa_vals, qm_vals, z_vals, sigma, n, beta, gamma, alpha=0, pi_rows=(), pi_values=()):
    """src/circuit.rs:442-489; the prover reaches it with no public inputs (src/prover.rs:224-227)."""
    gate_poly = trim(ntt(gate_values(a_col, b_col, c_col, qa_vals, qm_vals), n, inverse=True))
    perm_poly = trim(ntt(permutation_constraint_values(a_col, b_col, c_col, z_vals, sigma, n, beta, gamma), n, inverse=True))
    pub_poly = trim(ntt(public_input_values(a_col, n, alpha, pi_rows, pi_values), n, inverse=True))
    m = max(len(gate_poly), len(perm_poly), len(pub_poly))
    at = lambda p, i: p[i] if i < len(p) else 0
    t_num = trim([(at(gate_poly, i) + at(perm_poly, i) + at(pub_poly, i)) % R_MOD for i in range(m)])
    q, r = divide_by_vanishing(t_num, n)
    if r:
        raise NotDivisible("t(X) not divisible by Z_H(X)")
    return q


# ----------------------------------------------------------------------------- KZG10 (hiding bound 1) via marlin_pc
def kzg_commit(coeffs, srs_g, srs_gg, blind):
    """ark-poly-commit kzg10::commit: leading zero coefficients skipped, 3 blinding coefficients."""
    coeffs = trim(coeffs)
    C = msm(coeffs, srs_g[:len(coeffs)])
    return g1_add(C, msm(blind, srs_gg[:3]))


def kzg_open(coeffs, srs_g, srs_gg, blind, zeta, chal):
    """marlin_pc::open for one polynomial: opens chal*f with randomness chal*r.  Returns (W, random_v)."""
    p = trim([chal * c % R_MOD for c in coeffs])
    r = trim([chal * c % R_MOD for c in blind])
    w = div_linear(p, zeta)
    W = msm(w, srs_g[:len(w)])
    if r:
        hw = div_linear(r, zeta)
        W = g1_add(W, msm(hw, srs_gg[:len(hw)]))
    return W, poly_eval(r, zeta)


def kzg_check_with_trapdoor(C, W, v, random_v, zeta, chal, tau, gscal):
    """(tau - zeta) * W == chal*C - chal*v*G - random_v*gammaG, G1 only (needs the SRS trapdoor)."""
    lhs = g1_mul((tau - zeta) % R_MOD, W)
    rhs = g1_add(g1_mul(chal, C), g1_neg(g1_mul(chal * v % R_MOD, G1_GEN)))
    rhs = g1_add(rhs, g1_neg(g1_mul(random_v * gscal % R_MOD, G1_GEN)))
    return lhs == rhs


POLY_ORDER = ["a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma", "z", "t"]


def prove(inp, srs_g, srs_gg, blinds, chal, public_inputs=()):
    """src/prover.rs:119-336.  blinds: 27 Fr in draw order (3 per commit: a,b,c,q_add,q_mul,s_id,s_sigma,z,t).

    Returns dict with evals, comms, proofs (W, random_v), the four challenges and the
    canonical proof bytes (struct-field order of src/prover.rs:61-95)."""
    n = inp["n"]
    polys = {k: inp[k] for k in ["a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma"]}
    bl = {name: blinds[3 * i: 3 * i + 3] for i, name in enumerate(POLY_ORDER)}
    tr = Transcript(b"plonk_proof")
    if public_inputs:
        tr.append_message(b"public_inputs", b"".join(fr_bytes(x) for x in public_inputs))
    comms = {}
    for name in ["a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma"]:
        comms[name] = kzg_commit(polys[name], srs_g, srs_gg, bl[name])
    for name in ["q_add", "q_mul", "a", "b", "c", "s_id", "s_sigma"]:
        tr.append_message(name.encode(), commitment_bytes(comms[name]))
    beta = tr.challenge_scalar(b"beta")
    gamma = tr.challenge_scalar(b"gamma")
    _, z_coeffs = build_grand_product(inp["witness_flat"], inp["sigma"], n, beta, gamma)
    polys["z"] = z_coeffs
    comms["z"] = kzg_commit(z_coeffs, srs_g, srs_gg, bl["z"])
    tr.append_message(b"z", commitment_bytes(comms["z"]))
    alpha = tr.challenge_scalar(b"alpha")
    # src/prover.rs:228-237: columns are the coefficient vectors again; selectors and z go back to values
    pad = lambda v: list(v) + [0] * (n - len(v))
    a_col, b_col, c_col = polys["a"], polys["b"], polys["c"]
    if min(len(a_col), len(b_col), len(c_col)) < n:
        raise IndexError("wire column shorter than the domain (reference indexes out of bounds)")
    qa_vals = ntt(pad(polys["q_add"]), n)
    qm_vals = ntt(pad(polys["q_mul"]), n)
    z_vals = ntt(pad(z_coeffs), n)
    polys["t"] = build_quotient(a_col, b_col, c_col, qa_vals, qm_vals, z_vals, inp["sigma"], n, beta, gamma)
    comms["t"] = kzg_commit(polys["t"], srs_g, srs_gg, bl["t"])
    tr.append_message(b"t", commitment_bytes(comms["t"]))
    zeta = tr.challenge_scalar(b"zeta")
    evals = {name: poly_eval(polys[name], zeta) for name in POLY_ORDER}
    for name in POLY_ORDER:
        tr.append_message(b"eval_" + name.encode(), fr_bytes(evals[name]))
    proofs = {name: kzg_open(polys[name], srs_g, srs_gg, bl[name], zeta, chal) for name in POLY_ORDER}
    blob = b"".join(fr_bytes(evals[k]) for k in POLY_ORDER)
    blob += b"".join(commitment_bytes(comms[k]) for k in POLY_ORDER)
    blob += b"".join(opening_bytes(*proofs[k]) for k in POLY_ORDER)
    return dict(evals=evals, comms=comms, proofs=proofs, beta=beta, gamma=gamma, alpha=alpha, zeta=zeta,
                bytes=blob, polys=polys)
