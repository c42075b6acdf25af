"""Python face of the host prover (libplonk_host.so): the reference's prover.rs / poly_utils.rs surface on one MI355X.

    pk = ProverKey(device=0); pk.generate_srs(tau, gscal, length)
    inputs = pk.prepare(tables)                  # interpolation on the GPU + upload: src/main.rs:84,187-191
    proof = pk.prove(inputs, blinds, challenge)  # src/prover.rs:119-336, returns ProofResult

There is no CPU path: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
from dataclasses import dataclass

import numpy as np

from . import hip, synth

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(hip.LIB_DIR, "libplonk_host.so")
HOST_EXPORTS = ["plonk_host_last_error", "plonk_pk_create", "plonk_pk_destroy", "plonk_pk_ctx", "plonk_pk_load", "plonk_pk_generate", "plonk_pk_generate_cyclic",
                "plonk_pk_set_opening_challenge", "plonk_pk_set_sharding", "plonk_comm_get_unique_id", "plonk_pk_comm_init", "plonk_inputs_create", "plonk_inputs_destroy", "plonk_prove", "plonk_prove_host",
                "plonk_transcript_kat", "plonk_transcript_empty_beta", "plonk_serialize_g1", "plonk_verify", "plonk_pairing_selftest", "plonk_proof_roundtrip",
                "plonk_rng_create", "plonk_rng_destroy", "plonk_rng_next_u64", "plonk_rng_next_u32", "plonk_rng_fr", "plonk_rng_g1", "plonk_rng_g2",
                "plonk_chacha_block", "plonk_test_sponge_challenge", "plonk_pk_setup", "plonk_prove_rng", "plonk_verify_vk", "plonk_circuit_tables"]
GATHER_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64))

_host = None


def load_host_library():
    global _host
    if _host is None:
        hip.load_library()
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError(f"{HOST_LIB_PATH} is missing: build it with `make -C plonk_amd/host`")
        _host = C.CDLL(HOST_LIB_PATH)
        for name in HOST_EXPORTS:
            getattr(_host, name)
        _host.plonk_host_last_error.restype = C.c_char_p
        _host.plonk_pk_ctx.restype = C.c_void_p
        _host.plonk_pk_ctx.argtypes = [C.c_void_p]
        _host.plonk_rng_next_u64.restype = C.c_uint64
        _host.plonk_rng_next_u64.argtypes = [C.c_void_p]
        _host.plonk_rng_next_u32.restype = C.c_uint32
        _host.plonk_rng_next_u32.argtypes = [C.c_void_p]
    return _host


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(shape) if shape else a


@dataclass
class ProofResult:
    proof_bytes: bytes          # 1170 bytes: 9 evals, 9 commitments, 9 openings (field order of src/prover.rs:61-95)
    challenges: np.ndarray      # (4, 4): beta, gamma, alpha, zeta
    seconds: np.ndarray         # per-round host wall clock, last entry = total


class DeviceInputs:
    def __init__(self, pk, handle, log_n, polys):
        self.pk, self.h, self.log_n, self.polys = pk, handle, log_n, polys
        pk._inputs.add(self)          # the key closes its inputs before its context goes away (their buffers are freed through it)

    def close(self):
        if self.h:
            load_host_library().plonk_inputs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StdRng:
    """ark_std::test_rng() (src/main.rs:25) when seed is None, else rand 0.8 StdRng::from_seed(seed): ChaCha12."""

    def __init__(self, seed: bytes | None = None):
        self.lib = load_host_library()
        h = C.c_void_p()
        buf = None if seed is None else (C.c_uint8 * 32).from_buffer_copy(seed)
        if self.lib.plonk_rng_create(buf, C.byref(h)) != 0:
            raise RuntimeError("plonk_rng_create failed")
        self.h = h

    def __del__(self):
        try:
            if self.h:
                self.lib.plonk_rng_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def next_u64(self) -> int:
        return int(self.lib.plonk_rng_next_u64(self.h))

    def next_u32(self) -> int:
        return int(self.lib.plonk_rng_next_u32(self.h))

    def fr(self, n: int = 1) -> np.ndarray:
        """n draws of Fr::rand(rng), (n, 4) uint64"""
        out = np.zeros((n, 4), dtype=np.uint64)
        self.lib.plonk_rng_fr(self.h, C.c_uint64(n), _ptr(out))
        return out

    def g1(self) -> np.ndarray:
        out = np.zeros(8, dtype=np.uint64)
        self.lib.plonk_rng_g1(self.h, _ptr(out))
        return out

    def g2(self) -> np.ndarray:
        """x (c0, c1), y (c0, c1): 16 words"""
        out = np.zeros(16, dtype=np.uint64)
        self.lib.plonk_rng_g2(self.h, _ptr(out))
        return out


def circuit_tables(variables, gates):
    """The reference's CircuitBuilder / Circuit::from_builder / compute_sigma_mapping (src/circuit.rs) in the host library: same
    result as synth.circuit_tables, which builds the tables with numpy.  variables: integers; gates: (type, a, b, out)."""
    lib = load_host_library()
    ng = len(gates)
    n = 1
    while n < ng:
        n *= 2
    vals = synth.to_mont(variables)
    g = np.ascontiguousarray(np.array(gates, dtype=np.int64).reshape(-1, 4))
    cols = [np.zeros((n, 4), dtype=np.uint64) for _ in range(5)]
    flat = np.zeros((3 * n, 4), dtype=np.uint64)
    sigma = np.zeros(3 * n, dtype=np.uint64)
    rc = lib.plonk_circuit_tables(_ptr(vals), C.c_uint64(len(vals)), _ptr(g), C.c_uint64(ng), C.c_uint32(n.bit_length() - 1), *[_ptr(c) for c in cols],
                                  _ptr(flat), _ptr(sigma))
    if rc != 0:
        raise hip.PlonkHipError(rc, (lib.plonk_host_last_error() or b"").decode())
    return dict(n=n, log_n=n.bit_length() - 1, a=cols[0], b=cols[1], c=cols[2], q_add_vals=cols[3], q_mul_vals=cols[4], witness_flat=flat, sigma=sigma)


def chacha_block(key: bytes, counter: int, stream: int, rounds: int) -> np.ndarray:
    out = np.zeros(16, dtype=np.uint32)
    load_host_library().plonk_chacha_block((C.c_uint8 * 32).from_buffer_copy(key), C.c_uint64(counter), C.c_uint64(stream), C.c_int32(rounds), _ptr(out))
    return out


def test_sponge_challenge() -> np.ndarray:
    """The 128-bit value every PCS::open of the reference squeezes from test_sponge() (src/prover.rs:36-59,285-303), as an Fr."""
    out = np.zeros(4, dtype=np.uint64)
    load_host_library().plonk_test_sponge_challenge(_ptr(out))
    return out


test_sponge_challenge.__test__ = False     # not a pytest case


class ProverKey:
    """KZGProverKey (src/prover.rs:32) with its powers resident in HBM."""

    def __init__(self, device: int = 0):
        self.lib = load_host_library()
        h = C.c_void_p()
        self._ck(self.lib.plonk_pk_create(C.c_int32(device), C.byref(h)))
        self.h = h
        self._cb = None
        self._inputs = weakref.WeakSet()
        # the same plk_ctx seen through the raw C ABI
        self.ctx = hip.Context.__new__(hip.Context)
        self.ctx.lib = hip.load_library()
        self.ctx.h = C.c_void_p(self.lib.plonk_pk_ctx(self.h))

    def _ck(self, rc):
        if rc != 0:
            raise hip.PlonkHipError(rc, (self.lib.plonk_host_last_error() or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            for inp in list(self._inputs):     # device buffers of the inputs belong to this key's context: release them first
                inp.close()
            self.ctx.h = None
            self.lib.plonk_pk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_srs(self, g_xy, gamma_g_xy):
        g = _u64(g_xy, (-1, 8))
        self._ck(self.lib.plonk_pk_load(self.h, _ptr(g), C.c_uint64(len(g)), _ptr(_u64(gamma_g_xy, (-1, 8)))))

    def generate_srs(self, tau, gscal, length, shard_first=0, shard_len=None):
        sl = length if shard_len is None else shard_len
        self._ck(self.lib.plonk_pk_generate(self.h, _ptr(_u64(tau, (4,))), _ptr(_u64(gscal, (4,))), C.c_uint64(shard_first), C.c_uint64(sl), C.c_uint64(length)))

    def generate_srs_cyclic(self, tau, gscal, length, rank, world, log_block=12):
        """This rank's share of a block-cyclic SRS: block b of 2^log_block powers belongs to rank b mod world."""
        self._ck(self.lib.plonk_pk_generate_cyclic(self.h, _ptr(_u64(tau, (4,))), _ptr(_u64(gscal, (4,))), C.c_uint32(rank), C.c_uint32(world),
                                                   C.c_uint32(log_block), C.c_uint64(length)))

    def setup(self, max_degree: int, rng: StdRng, rank: int = 0, world: int = 1, log_block: int = 12) -> np.ndarray:
        """PCS::setup + PCS::trim (src/main.rs:182-183): draws beta, g, gamma_g, h from rng, builds the max_degree + 1 powers on the GPU.
        Returns the verifier key as 48 words: g, gamma_g, h (x, y), beta_h (x, y)."""
        vk = np.zeros(48, dtype=np.uint64)
        self._ck(self.lib.plonk_pk_setup(self.h, C.c_uint64(max_degree), rng.h, C.c_uint32(rank), C.c_uint32(world), C.c_uint32(log_block), _ptr(vk)))
        return vk

    def set_opening_challenge(self, chal):
        self._ck(self.lib.plonk_pk_set_opening_challenge(self.h, _ptr(_u64(chal, (4,)))))

    def set_sharding(self, rank, world_size, gather):
        """gather(mine: np.ndarray(count, 16) uint64) -> np.ndarray(world_size, count, 16): the partial points of all ranks.
        Called three times per proof (count = 7, 2, 9)."""
        def cb(_user, mine, count, out):
            m = np.ctypeslib.as_array(mine, shape=(count, 16)).copy()
            res = np.ascontiguousarray(gather(m), dtype=np.uint64).reshape(world_size * count * 16)
            C.memmove(out, res.ctypes.data, world_size * count * 16 * 8)
        self._cb = GATHER_FN(cb)
        self._ck(self.lib.plonk_pk_set_sharding(self.h, C.c_uint32(rank), C.c_uint32(world_size), self._cb, None))

    def comm_unique_id(self) -> bytes:
        """128 bytes that rank 0 hands to every rank (ncclGetUniqueId)"""
        buf = (C.c_uint8 * 128)()
        self._ck(self.lib.plonk_comm_get_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        """set_sharding with the all-gather inside the library (RCCL), no Python in the data path"""
        self._ck(self.lib.plonk_pk_comm_init(self.h, (C.c_uint8 * 128).from_buffer_copy(unique_id), C.c_uint32(rank), C.c_uint32(world_size)))

    def prepare(self, tables) -> DeviceInputs:
        """src/main.rs:79-84,187-191: interpolates q_add, q_mul, s_id, s_sigma on the GPU and uploads everything."""
        n, log_n = tables["n"], tables["log_n"]
        log_n4 = (3 * n - 1).bit_length()
        q_add = synth.trim(self.ctx.interpolate_selector(tables["q_add_vals"]))
        q_mul = synth.trim(self.ctx.interpolate_selector(tables["q_mul_vals"]))
        s_id, s_sigma = self.ctx.interpolate_permutation(tables["sigma"], log_n4)
        polys = [synth.trim(tables["a"]), synth.trim(tables["b"]), synth.trim(tables["c"]), q_add, q_mul, synth.trim(s_id), synth.trim(s_sigma)]
        return self.upload(polys, tables["witness_flat"], tables["sigma"], log_n)

    def upload(self, polys, witness_flat, sigma, log_n) -> DeviceInputs:
        ps = [_u64(p, (-1, 4)) for p in polys]
        ptrs = (C.c_void_p * 7)(*[p.ctypes.data if len(p) else None for p in ps])
        lens = np.array([len(p) for p in ps], dtype=np.uint64)
        h = C.c_void_p()
        self._ck(self.lib.plonk_inputs_create(self.h, ptrs, _ptr(lens), _ptr(_u64(witness_flat)), _ptr(_u64(sigma)), C.c_uint32(log_n), C.byref(h)))
        return DeviceInputs(self, h, log_n, ps)

    def prove(self, inputs: DeviceInputs, blinds, public_inputs=None) -> ProofResult:
        bl = _u64(blinds, (-1, 4))
        pub = _u64(public_inputs, (-1, 4)) if public_inputs is not None and len(public_inputs) else None
        proof = np.zeros(1170, dtype=np.uint8); ch = np.zeros((4, 4), dtype=np.uint64); sec = np.zeros(6, dtype=np.float64)
        self._ck(self.lib.plonk_prove(self.h, inputs.h, _ptr(pub), C.c_uint64(0 if pub is None else len(pub)), _ptr(bl), C.c_uint64(len(bl)), _ptr(proof),
                                      _ptr(ch), _ptr(sec)))
        return ProofResult(proof.tobytes(), ch, sec)


    def prove_host(self, polys, witness_flat, sigma, log_n, blinds, public_inputs=None) -> ProofResult:
        """create_plonk_proof_with_transcript on host arrays (src/prover.rs:119-133 borrows host vectors): a, b, c, q_add, q_mul, s_id,
        s_sigma as (len, 4) uint64 arrays, not copied; they cross PCIe while the first commitments run (at link speed from arrays made
        with ctx.host_empty, staged by the runtime from ordinary ones)."""
        ps = [p if (p.dtype == np.uint64 and p.flags["C_CONTIGUOUS"]) else _u64(p, (-1, 4)) for p in polys]
        w = witness_flat if (witness_flat.dtype == np.uint64 and witness_flat.flags["C_CONTIGUOUS"]) else _u64(witness_flat)
        sg = sigma if (sigma.dtype == np.uint64 and sigma.flags["C_CONTIGUOUS"]) else _u64(sigma)
        ptrs = (C.c_void_p * 7)(*[p.ctypes.data if len(p) else None for p in ps])
        lens = np.array([len(p) for p in ps], dtype=np.uint64)
        bl = _u64(blinds, (-1, 4))
        pub = _u64(public_inputs, (-1, 4)) if public_inputs is not None and len(public_inputs) else None
        proof = np.zeros(1170, dtype=np.uint8); ch = np.zeros((4, 4), dtype=np.uint64); sec = np.zeros(6, dtype=np.float64)
        self._ck(self.lib.plonk_prove_host(self.h, ptrs, _ptr(lens), _ptr(w), _ptr(sg), C.c_uint32(log_n), _ptr(pub), C.c_uint64(0 if pub is None else len(pub)),
                                           _ptr(bl), C.c_uint64(len(bl)), _ptr(proof), _ptr(ch), _ptr(sec)))
        return ProofResult(proof.tobytes(), ch, sec)

    def prove_rng(self, inputs: DeviceInputs, rng: StdRng, public_inputs=None) -> ProofResult:
        """prove() with the 27 blinding coefficients drawn from rng (the reference's `rng: &mut R`, src/prover.rs:132)."""
        pub = _u64(public_inputs, (-1, 4)) if public_inputs is not None and len(public_inputs) else None
        proof = np.zeros(1170, dtype=np.uint8); ch = np.zeros((4, 4), dtype=np.uint64); sec = np.zeros(6, dtype=np.float64)
        self._ck(self.lib.plonk_prove_rng(self.h, inputs.h, _ptr(pub), C.c_uint64(0 if pub is None else len(pub)), rng.h, _ptr(proof), _ptr(ch), _ptr(sec)))
        return ProofResult(proof.tobytes(), ch, sec)


def verify_vk(vk, proof_bytes: bytes, opening_challenge, public_inputs=None) -> bool:
    """verify() with an explicit verifier key (the 48 words ProverKey.setup returns)."""
    lib = load_host_library()
    buf = np.frombuffer(proof_bytes, dtype=np.uint8).copy()
    pub = _u64(public_inputs, (-1, 4)) if public_inputs is not None and len(public_inputs) else None
    ok = C.c_int32(0)
    rc = lib.plonk_verify_vk(_ptr(_u64(vk, (48,))), _ptr(buf), C.c_uint64(len(buf)), _ptr(pub), C.c_uint64(0 if pub is None else len(pub)),
                             _ptr(_u64(opening_challenge, (4,))), C.byref(ok))
    if rc != 0:
        raise hip.PlonkHipError(rc, (lib.plonk_host_last_error() or b"").decode())
    return bool(ok.value)


def verify(tau, gscal, proof_bytes: bytes, opening_challenge, public_inputs=None) -> bool:
    """verify_plonk_proof_with_transcript (src/verifier.rs:57) on the CPU: transcript replay and nine KZG10 checks with pairings.
    tau, gscal: trapdoor of the synthetic SRS the proof was made with (the verifier key is derived from them)."""
    lib = load_host_library()
    buf = np.frombuffer(proof_bytes, dtype=np.uint8).copy()
    pub = _u64(public_inputs, (-1, 4)) if public_inputs is not None and len(public_inputs) else None
    ok = C.c_int32(0)
    rc = lib.plonk_verify(_ptr(_u64(tau, (4,))), _ptr(_u64(gscal, (4,))), _ptr(buf), C.c_uint64(len(buf)), _ptr(pub), C.c_uint64(0 if pub is None else len(pub)),
                          _ptr(_u64(opening_challenge, (4,))), C.byref(ok))
    if rc != 0:
        raise hip.PlonkHipError(rc, (lib.plonk_host_last_error() or b"").decode())
    return bool(ok.value)
