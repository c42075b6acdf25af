"""Synthetic prover inputs (this build's own generator; the reference only ships a 4-gate example, src/main.rs:29-39).

Random arithmetic circuit with exactly 2^log_n gates: v0, v1 uniform in Fr; gate 0 uses (v0, v1); gate i > 0 takes the
previous gate's output as its left input and a uniformly chosen earlier variable as its right input; the type is Add or
Mul with probability 1/2 (SplitMix64 stream, seed given).  The last row is re-drawn until none of its three values is zero
(the reference indexes past the end of a trimmed wire column otherwise, src/prover.rs:229-231).  Columns, witness_flat and
sigma follow src/circuit.rs:216-276 and :80-112.
"""
from __future__ import annotations

import numpy as np

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_M64 = (1 << 64) - 1
MONT_R = (1 << 256) % R_MOD


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
        while True:
            v = 0
            for i in range(4):
                v |= self.next_u64() << (64 * i)
            v &= (1 << 254) - 1
            if v < R_MOD:
                return v


def to_mont(vals) -> np.ndarray:
    """Python ints -> (n, 4) uint64 limbs of x * 2^256 mod r."""
    buf = b"".join(((int(v) * MONT_R) % R_MOD).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def from_mont(arr) -> list:
    rinv = pow(MONT_R, -1, R_MOD)
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") * rinv % R_MOD for i in range(0, len(b), 32)]


def example_circuit():
    """src/main.rs:29-39."""
    variables = [3, 4, 7, 2, 14, 5, 19, 57]
    gates = [(0, 0, 1, 2), (1, 2, 3, 4), (0, 4, 5, 6), (1, 6, 0, 7)]   # (type: 0 Add / 1 Mul, a, b, out)
    return variables, gates


def synthetic_circuit(log_n: int, seed: int = 1):
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
            va, vb = variables[a], variables[b]
            v = (va + vb) % R_MOD if t == 0 else va * vb % R_MOD
            if i == n - 1 and (va == 0 or vb == 0 or v == 0):
                continue
            break
        variables.append(v)
        gates.append((t, a, b, len(variables) - 1))
    return variables, gates


def circuit_tables(variables, gates):
    """Columns (Montgomery limbs, zero padded to the domain), witness_flat, sigma: the driver's steps src/main.rs:42-100."""
    ng = len(gates)
    n = 1
    while n < ng:
        n *= 2
    g = np.array(gates, dtype=np.int64).reshape(-1, 4)
    zero = [0] * (n - ng)
    a_vals = [variables[i] for i in g[:, 1]] + zero
    b_vals = [variables[i] for i in g[:, 2]] + zero
    c_vals = [variables[i] for i in g[:, 3]] + zero
    cols = [to_mont(a_vals), to_mont(b_vals), to_mont(c_vals)]
    one = to_mont([1])[0]
    q_add = np.zeros((n, 4), dtype=np.uint64)
    q_mul = np.zeros((n, 4), dtype=np.uint64)
    q_add[:ng][g[:, 0] == 0] = one
    q_mul[:ng][g[:, 0] == 1] = one
    witness_flat = np.stack(cols, axis=1).reshape(3 * n, 4)
    # sigma: every variable's uses (flat index 3*row + col, in row order) form one cycle
    var_at = np.full(3 * n, -1, dtype=np.int64)
    var_at[0:3 * ng:3] = g[:, 1]
    var_at[1:3 * ng:3] = g[:, 2]
    var_at[2:3 * ng:3] = g[:, 3]
    used = np.nonzero(var_at >= 0)[0]
    order = used[np.argsort(var_at[used], kind="stable")]
    grp = var_at[order]
    nxt = np.roll(order, -1)
    first_of_group = np.ones(len(order), dtype=bool)
    first_of_group[1:] = grp[1:] != grp[:-1]
    starts = np.nonzero(first_of_group)[0]
    last = np.r_[starts[1:] - 1, len(order) - 1]
    nxt[last] = order[starts]
    sigma = np.arange(3 * n, dtype=np.uint64)
    sigma[order] = nxt.astype(np.uint64)
    return dict(n=n, log_n=n.bit_length() - 1, a=cols[0], b=cols[1], c=cols[2], q_add_vals=q_add, q_mul_vals=q_mul, witness_flat=witness_flat, sigma=sigma)


def trim(arr: np.ndarray) -> np.ndarray:
    """DensePolynomial::from_coefficients_vec: strip trailing zero coefficients."""
    nz = np.nonzero(arr.any(axis=1))[0]
    return arr[: (nz[-1] + 1 if len(nz) else 0)]


def seeded_scalars(count: int, seed: int) -> np.ndarray:
    rng = SplitMix64(seed)
    return to_mont([rng.next_fr() for _ in range(count)])


def cyclic_log_block(log_gates: int) -> int:
    """Block size (as a power of two) of the block-cyclic SRS shards for a proof of 2^log_gates gates: a multiple of the block in which
    the opening divisions walk their polynomials (2048 coefficients below 2^21 coefficients, 8192 from there on), so that a rank's
    share of the quotient coefficients lies in whole division blocks and the division writes out nothing that another rank uses."""
    return 13 if log_gates >= 21 else 12
