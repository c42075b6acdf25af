"""ctypes wrapper around oracle/_build/liboracle.so (TEST INFRASTRUCTURE - never imported by plonk_amd).

Arrays are numpy uint64 of shape (n, 4) [field elements, Montgomery form] or (n, 8) [affine x||y].
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_prove.restype = C.c_int
        _lib.orc_quotient.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


# ---------------------------------------------------------------- int <-> limb helpers (host side, Python ints)
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def ints_to_canonical(vals) -> np.ndarray:
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def canonical_to_ints(arr) -> list:
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def fr_from_ints(vals) -> np.ndarray:
    """Python ints (canonical) -> (n,4) Montgomery limbs."""
    c = ints_to_canonical([v % R_MOD for v in vals])
    out = np.empty_like(c)
    if len(c):
        lib().orc_fr_from_canonical(_p(c), _p(out), C.c_uint64(len(c)))
    return out


def fr_to_ints(arr) -> list:
    a = _u64(arr).reshape(-1, 4)
    out = np.empty_like(a)
    if len(a):
        lib().orc_fr_to_canonical(_p(a), _p(out), C.c_uint64(len(a)))
    return canonical_to_ints(out)


def fq_from_ints(vals) -> np.ndarray:
    c = ints_to_canonical([v % Q_MOD for v in vals])
    out = np.empty_like(c)
    if len(c):
        lib().orc_fq_from_canonical(_p(c), _p(out), C.c_uint64(len(c)))
    return out


def fq_to_ints(arr) -> list:
    a = _u64(arr).reshape(-1, 4)
    out = np.empty_like(a)
    if len(a):
        lib().orc_fq_to_canonical(_p(a), _p(out), C.c_uint64(len(a)))
    return canonical_to_ints(out)


def points_from_ints(pts) -> np.ndarray:
    """[(x, y)] -> (n, 8) Montgomery limbs (no infinity)."""
    flat = []
    for x, y in pts:
        flat += [x, y]
    return fq_from_ints(flat).reshape(-1, 8)


def point_to_ints(xy, inf):
    if inf:
        return None
    v = fq_to_ints(np.asarray(xy, dtype=np.uint64).reshape(2, 4))
    return (v[0], v[1])


# ---------------------------------------------------------------- operations
OPS = {"add": 0, "sub": 1, "mul": 2, "inv": 3, "neg": 4}


def set_threads(t: int) -> int:
    """threads of the MSM (1 = the reference's single-threaded arkworks build); returns the previous setting"""
    return int(lib().orc_set_threads(C.c_int(t)))


def fr_op(op, a, b=None):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_fr_op(OPS[op], _p(a), _p(_u64(b)) if b is not None else None, _p(out), C.c_uint64(a.size // 4))
    return out


def fq_op(op, a, b=None):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_fq_op(OPS[op], _p(a), _p(_u64(b)) if b is not None else None, _p(out), C.c_uint64(a.size // 4))
    return out


def g1_generator():
    out = np.empty(8, dtype=np.uint64); lib().orc_g1_generator(_p(out)); return out


def g1_add(p, pinf, q, qinf):
    out = np.empty(8, dtype=np.uint64); oinf = C.c_int32(0)
    lib().orc_g1_add(_p(_u64(p)), int(pinf), _p(_u64(q)), int(qinf), _p(out), C.byref(oinf))
    return out, oinf.value


def g1_mul(p, pinf, scalar):
    out = np.empty(8, dtype=np.uint64); oinf = C.c_int32(0)
    lib().orc_g1_mul(_p(_u64(p)), int(pinf), _p(_u64(scalar)), _p(out), C.byref(oinf))
    return out, oinf.value


def g1_compress(p, pinf) -> bytes:
    out = np.empty(32, dtype=np.uint8); lib().orc_g1_compress(_p(_u64(p)), int(pinf), _p(out)); return out.tobytes()


def srs_powers(tau, scale, first, count):
    """scale * tau^i * G for i in [first, first+count) as (count, 8) limbs."""
    out = np.empty((count, 8), dtype=np.uint64)
    lib().orc_srs_powers(_p(_u64(tau)), _p(_u64(scale)), C.c_uint64(first), C.c_uint64(count), _p(out))
    return out


def ntt(data, inverse=False, coset=None):
    a = _u64(data).reshape(-1, 4).copy()
    log_n = (len(a) - 1).bit_length()
    assert 1 << log_n == len(a)
    lib().orc_ntt(_p(a), log_n, int(inverse), _p(_u64(coset)) if coset is not None else None)
    return a


def msm(scalars, bases):
    s = _u64(scalars).reshape(-1, 4); b = _u64(bases).reshape(-1, 8)
    out = np.empty(8, dtype=np.uint64); oinf = C.c_int32(0)
    lib().orc_msm(_p(s), _p(b), C.c_uint64(len(s)), _p(out), C.byref(oinf))
    return out, oinf.value


def poly_eval(coeffs, x):
    c = _u64(coeffs).reshape(-1, 4); out = np.empty(4, dtype=np.uint64)
    lib().orc_poly_eval(_p(c), C.c_uint64(len(c)), _p(_u64(x)), _p(out))
    return out


def div_linear(coeffs, z):
    c = _u64(coeffs).reshape(-1, 4); out = np.zeros((max(len(c) - 1, 0), 4), dtype=np.uint64)
    lib().orc_div_linear(_p(c), C.c_uint64(len(c)), _p(_u64(z)), _p(out))
    return out


def grand_product(witness_flat, sigma, beta, gamma):
    w = _u64(witness_flat).reshape(-1, 4); s = _u64(sigma)
    n = len(w) // 3; log_n = (n - 1).bit_length()
    zv = np.empty((n, 4), dtype=np.uint64); zc = np.empty((n, 4), dtype=np.uint64)
    lib().orc_grand_product(_p(w), _p(s), log_n, _p(_u64(beta)), _p(_u64(gamma)), _p(zv), _p(zc))
    return zv, zc


def quotient(a, b, c, qa, qm, zv, sigma, beta, gamma) -> int:
    a = _u64(a).reshape(-1, 4); n = len(a); log_n = (n - 1).bit_length()
    return lib().orc_quotient(_p(a), _p(_u64(b)), _p(_u64(c)), _p(_u64(qa)), _p(_u64(qm)), _p(_u64(zv)), _p(_u64(sigma)), log_n,
                              _p(_u64(beta)), _p(_u64(gamma)))


def quotient_pi(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, pi_rows, pi_values) -> int:
    """build_quotient_polynomial with the public-input polynomial of src/circuit.rs:410-439: 0 divisible, 1 not, 2 row outside the domain"""
    a = _u64(a).reshape(-1, 4); n = len(a); log_n = (n - 1).bit_length()
    rows = np.ascontiguousarray(pi_rows, dtype=np.uint64); vals = _u64(pi_values).reshape(-1, 4)
    assert len(rows) == len(vals)
    return lib().orc_quotient_pi(_p(a), _p(_u64(b)), _p(_u64(c)), _p(_u64(qa)), _p(_u64(qm)), _p(_u64(zv)), _p(_u64(sigma)), log_n,
                                 _p(_u64(beta)), _p(_u64(gamma)), _p(_u64(alpha)), _p(rows), _p(vals), C.c_uint64(len(rows)))


def quotient_coset(a, b, c, qa, qm, zv, sigma, beta, gamma, alpha):
    """(status, 4n coefficients of t) of the extended quotient, computed without a coset (see orc_quotient_coset)."""
    a = _u64(a).reshape(-1, 4); n = len(a); log_n = (n - 1).bit_length()
    out = np.zeros((4 * n, 4), dtype=np.uint64)
    rc = lib().orc_quotient_coset(_p(a), _p(_u64(b)), _p(_u64(c)), _p(_u64(qa)), _p(_u64(qm)), _p(_u64(zv)), _p(_u64(sigma)), log_n,
                                  _p(_u64(beta)), _p(_u64(gamma)), _p(_u64(alpha)), _p(out))
    return rc, out


def kzg_commit(g, gg, coeffs, blind):
    g = _u64(g).reshape(-1, 8); c = _u64(coeffs).reshape(-1, 4)
    out = np.empty(8, dtype=np.uint64); oinf = C.c_int32(0)
    lib().orc_kzg_commit(_p(g), C.c_uint64(len(g)), _p(_u64(gg)), _p(c), C.c_uint64(len(c)), _p(_u64(blind)), _p(out), C.byref(oinf))
    return out, oinf.value


def kzg_open(g, gg, coeffs, blind, zeta, chal):
    g = _u64(g).reshape(-1, 8); c = _u64(coeffs).reshape(-1, 4)
    out = np.empty(8, dtype=np.uint64); oinf = C.c_int32(0); rv = np.empty(4, dtype=np.uint64)
    lib().orc_kzg_open(_p(g), C.c_uint64(len(g)), _p(_u64(gg)), _p(c), C.c_uint64(len(c)), _p(_u64(blind)), _p(_u64(zeta)), _p(_u64(chal)),
                       _p(out), C.byref(oinf), _p(rv))
    return out, oinf.value, rv


def transcript_kat() -> bytes:
    out = np.empty(32, dtype=np.uint8); lib().orc_transcript_kat(_p(out)); return out.tobytes()


def empty_beta():
    out = np.empty(4, dtype=np.uint64); lib().orc_empty_beta(_p(out)); return out


def prove(g, gg, polys, witness_flat, sigma, log_n, blinds, chal, public_inputs=None):
    """polys: list of 7 (len,4) arrays in the order a,b,c,q_add,q_mul,s_id,s_sigma.

    Returns (status, proof_bytes, challenges (4,4), stage_seconds[6])."""
    g = _u64(g).reshape(-1, 8)
    ps = [_u64(p).reshape(-1, 4) for p in polys]
    ptrs = (C.c_void_p * 7)(*[p.ctypes.data for p in ps])
    lens = np.array([len(p) for p in ps], dtype=np.uint64)
    pub = _u64(public_inputs).reshape(-1, 4) if public_inputs is not None and len(public_inputs) else None
    proof = np.zeros(1170, dtype=np.uint8); ch = np.zeros((4, 4), dtype=np.uint64); st = np.zeros(6, dtype=np.float64)
    rc = lib().orc_prove(_p(g), C.c_uint64(len(g)), _p(_u64(gg)), ptrs, _p(lens), _p(pub), C.c_uint64(0 if pub is None else len(pub)),
                         _p(_u64(witness_flat)), _p(_u64(sigma)), int(log_n), _p(_u64(blinds)), _p(_u64(chal)), _p(proof), _p(ch), _p(st))
    return rc, proof.tobytes(), ch, st
