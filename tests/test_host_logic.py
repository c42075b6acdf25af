"""CPU suite, part 2: product code that runs without a GPU.

 - the device field / curve headers (fp29.cuh, g1.cuh) compiled for the host, against big-integer arithmetic;
 - the host prover's transcript and serialization (libplonk_host.so) against the external anchors;
 - the C-ABI libraries load and export every symbol include/plonk_hip.h declares;
 - the synthetic-input generator against the model;
 - the sharded-MSM combine step over a 2-rank gloo group.
"""
import ctypes as C
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import pymodel as pm
from plonk_amd import hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fp29():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native"), "-s"])
    return C.CDLL(os.path.join(ROOT, "tests", "native", "libfp29_host.so"))


def _words(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint32).copy()


def _val(w):
    return int.from_bytes(w.tobytes(), "little")


def _call(fn, what, a, b=None):
    out = np.zeros(8, np.uint32); A = _words(a); B = _words(b) if b is not None else None
    fn(what, A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p) if B is not None else None, out.ctypes.data_as(C.c_void_p))
    return _val(out)


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_fp29_arithmetic(fp29, field):
    fn, p = (fp29.fp29_fr, pm.R_MOD) if field == "fr" else (fp29.fp29_fq, pm.Q_MOD)
    ri = pow(1 << 261, -1, p)
    rnd = random.Random(5)
    vals = [0, 1, 2, p - 1, p - 2, 1 << 253, (1 << 232) - 1, 1 << 232, (1 << 29) - 1, 1 << 29] + [rnd.randrange(p) for _ in range(400)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert _call(fn, 0, a, b) == a * b * ri % p
        assert _call(fn, 1, a) == a * a * ri % p
        assert _call(fn, 8, a, b) == (a * (a - b) - b * b) * ri % p     # two products, one reduction, lazily reduced operands
        assert _call(fn, 2, a, b) == (a + b) % p
        assert _call(fn, 3, a, b) == (a - b) % p
        assert _call(fn, 7, a) == a                                   # unpack / pack round trip
        assert _call(fn, 5, a, b) == (1 if a == b else 0)             # exact zero test on lazily reduced values
        assert _call(fn, 5, a, a) == 1 and _call(fn, 6, a, a) == 1
        assert _call(fn, 9, a, b) == 2 * a * b % p                    # fixed-operand (Shoup) product, lazily reduced x of bound 18
        if b:
            assert _call(fn, 10, a, b) == (-a * b) % p                # ... with the negated operand derived from the pair
    for x in vals[10:40] + [1, p - 1]:
        assert _call(fn, 4, x * (1 << 261) % p) == pow(x, -1, p) * (1 << 261) % p


def test_xyzz_group_law(fp29):
    q = pm.Q_MOD
    R256 = 1 << 256
    rnd = random.Random(6)

    def pt_words(P):
        return np.concatenate([_words(P[0] * R256 % q), _words(P[1] * R256 % q)])

    def from_xyzz(out):
        ri = pow(R256, -1, q)
        X, Y, ZZ, ZZZ = [_val(out[8 * i:8 * i + 8]) * ri % q for i in range(4)]
        if ZZ == 0:
            return None
        assert pow(ZZ, 3, q) == pow(ZZZ, 2, q)
        return (X * pow(ZZ, -1, q) % q, Y * pow(ZZZ, -1, q) % q)

    def chain(ps, neg):
        arr = np.concatenate([pt_words(P) for P in ps]); out = np.zeros(32, np.uint32)
        fp29.g1_madd_chain(arr.ctypes.data_as(C.c_void_p), len(ps), neg, out.ctypes.data_as(C.c_void_p))
        return from_xyzz(out)

    def tree(ps, d):
        arr = np.concatenate([pt_words(P) for P in ps]); out = np.zeros(32, np.uint32)
        fp29.g1_add_tree(arr.ctypes.data_as(C.c_void_p), len(ps), d, out.ctypes.data_as(C.c_void_p))
        return from_xyzz(out)

    def ref_sum(ps, neg=0):
        acc = None
        for i, P in enumerate(ps):
            acc = pm.g1_add(acc, pm.g1_neg(P) if (neg and i & 1) else P)
        return acc

    G = pm.G1_GEN
    pts = [pm.g1_mul(rnd.randrange(1, pm.R_MOD), G) for _ in range(10)]
    for n in range(1, 11):
        assert chain(pts[:n], 0) == ref_sum(pts[:n]) and chain(pts[:n], 1) == ref_sum(pts[:n], 1)
        assert tree(pts[:n], 0) == ref_sum(pts[:n]) and tree(pts[:n], 3) == pm.g1_mul(8, ref_sum(pts[:n]))
    P, Q = pts[0], pts[1]
    assert chain([P, P], 0) == pm.g1_mul(2, P) and chain([P, P], 1) is None          # doubling and cancellation inside madd
    assert chain([P] * 5, 0) == pm.g1_mul(5, P) and chain([P, P, Q], 1) == Q
    assert tree([P, Q, P, Q], 0) == pm.g1_mul(2, pm.g1_add(P, Q))                     # doubling inside the full addition
    assert tree([P, Q, pm.g1_neg(P), pm.g1_neg(Q)], 0) is None
    assert chain([G] * 7, 0) == pm.g1_mul(7, G)


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(plk_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 40
    assert sorted(hip.EXPORTS) == declared
    lib = hip.load_library()
    for name in declared:
        assert hasattr(lib, name), name


def test_host_library_exports_every_declared_symbol():
    """include/plonk_host.h (the whole-proof surface the Rust drop-in binds) against libplonk_host.so"""
    from plonk_amd import prover
    header = open(os.path.join(ROOT, "include", "plonk_host.h")).read()
    declared = sorted(set(re.findall(r"\b(plonk_[a-z0-9_]+)\s*\(", header)) - {"plonk_gather_fn"})
    assert len(declared) >= 20 and "plonk_prove_host" in declared
    lib = prover.load_host_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) <= set(prover.HOST_EXPORTS) | {"plonk_rng_next_u64", "plonk_rng_next_u32"}


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(hip.PlonkHipError) as e:
        hip.Context(0)
    assert e.value.status in ("PLK_ERR_NO_DEVICE", "PLK_ERR_HIP")


def test_host_transcript_and_serialization():
    from plonk_amd import prover
    lib = prover.load_host_library()
    out = np.zeros(32, np.uint8)
    lib.plonk_transcript_kat(out.ctypes.data_as(C.c_void_p))
    assert out.tobytes().hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    beta = np.zeros(4, np.uint64)
    lib.plonk_transcript_empty_beta(beta.ctypes.data_as(C.c_void_p))
    assert synth.from_mont(beta)[0] == 17135754953240904778603612952250948273532778453732849488215102589457823141935
    rnd = random.Random(3)
    for k in [1, 2, 3] + [rnd.randrange(1, pm.R_MOD) for _ in range(20)]:
        P = pm.g1_mul(k, pm.G1_GEN)
        xy = np.concatenate([np.frombuffer((c * (1 << 256) % pm.Q_MOD).to_bytes(32, "little"), dtype=np.uint64) for c in P])
        lib.plonk_serialize_g1(xy.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p))
        assert out.tobytes() == pm.g1_compressed(P)
    lib.plonk_serialize_g1(np.zeros(8, np.uint64).ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p))
    assert out.tobytes() == pm.g1_compressed(None)


def test_synthetic_generator_matches_model():
    for log_n in (1, 2, 3, 6):
        v, g = pm.synthetic_circuit(log_n, seed=log_n)
        v2, g2 = synth.synthetic_circuit(log_n, seed=log_n)
        assert v == v2 and g == g2
        inp = pm.prepare_inputs(v, g)
        tb = synth.circuit_tables(v2, g2)
        assert list(tb["sigma"]) == inp["sigma"]
        assert synth.from_mont(tb["witness_flat"]) == inp["witness_flat"]
        assert synth.from_mont(tb["q_add_vals"]) == inp["cols"][3] and synth.from_mont(tb["q_mul_vals"]) == inp["cols"][4]
    tb = synth.circuit_tables(*synth.example_circuit())
    assert list(tb["sigma"]) == [10, 1, 3, 2, 4, 6, 5, 7, 9, 8, 0, 11]


def test_sum_partials_host_combine(orc):
    """plk_g1_sum_partials (pure host code): adds per-shard partial points and normalises."""
    lib = hip.load_library()
    one_q = orc.fq_from_ints([1])[0]
    rnd = random.Random(9)
    pts = [pm.g1_mul(rnd.randrange(1, pm.R_MOD), pm.G1_GEN) for _ in range(5)]
    parts = np.zeros((7, 16), dtype=np.uint64)
    for i, P in enumerate(pts):
        parts[i, :8] = orc.points_from_ints([P])[0]; parts[i, 8:12] = one_q; parts[i, 12:16] = one_q
    parts[5] = 0                                            # identity
    parts[6] = parts[0]                                     # a repeated point forces the doubling branch
    out = np.zeros(8, np.uint64); inf = C.c_int32(0)
    assert lib.plk_g1_sum_partials(parts.ctypes.data_as(C.c_void_p), C.c_uint64(7), out.ctypes.data_as(C.c_void_p), C.byref(inf)) == 0
    want = None
    for P in pts + [pts[0]]:
        want = pm.g1_add(want, P)
    assert orc.point_to_ints(out, inf.value) == want


WORKER = r'''
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from oracle import oracle as orc, pymodel as pm
from plonk_amd import hip
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
rng = pm.SplitMix64(77)
n = 37
tau = orc.fr_from_ints([rng.next_fr()])[0]
g = orc.srs_powers(tau, orc.fr_from_ints([1])[0], 0, n)
sc = orc.fr_from_ints([rng.next_fr() for _ in range(n)])
per = (n + world - 1) // world
lo, hi = rank * per, min(n, rank * per + per)
part, inf = orc.msm(sc[lo:hi], g[lo:hi])                       # this rank's point-range shard
mine = np.zeros(16, np.uint64)
if not inf:
    mine[:8] = part; mine[8:12] = orc.fq_from_ints([1])[0]; mine[12:16] = orc.fq_from_ints([1])[0]
recv = [torch.zeros(16, dtype=torch.int64) for _ in range(world)]
dist.all_gather(recv, torch.from_numpy(mine.view(np.int64)))
allp = np.stack([r.numpy().view(np.uint64) for r in recv])
lib = hip.load_library()
out = np.zeros(8, np.uint64); oinf = C.c_int32(0)
assert lib.plk_g1_sum_partials(allp.ctypes.data_as(C.c_void_p), C.c_uint64(world), out.ctypes.data_as(C.c_void_p), C.byref(oinf)) == 0
full, finf = orc.msm(sc, g)
assert oinf.value == finf and np.array_equal(out, full), "sharded sum differs from the full MSM"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_point_range_shards_combine_over_gloo(tmp_path):
    """World size 2 on the CPU: per-rank partial sums, all_gather, host combine == unsharded MSM."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_pairing_is_bilinear():
    """The host verifier's ate pairing: e(aG, bH) = e(G, H)^(ab) and e(G, H) != 1."""
    from plonk_amd import prover
    lib = prover.load_host_library()
    for a, b in ((1, 1), (2, 3), (pm.R_MOD - 1, 5), (0x123456789ABCDEF, pm.R_MOD - 2)):
        am, bm = synth.to_mont([a])[0], synth.to_mont([b])[0]
        assert lib.plonk_pairing_selftest(am.ctypes.data_as(C.c_void_p), bm.ctypes.data_as(C.c_void_p)) == 1


def test_example_circuit_proved_and_verified_on_the_cpu(orc):
    """BASELINE configs[0]: the reference's example circuit (src/main.rs:29-39) proved by the CPU oracle, then verified by the host
    verifier (src/verifier.rs:57-236: transcript replay + nine KZG10 pairing checks), no GPU and no trapdoor shortcut involved."""
    import json
    from plonk_amd import prover
    p = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))[0]
    proof = bytes.fromhex(p["proof_hex"])
    tau, gs, chal = (synth.to_mont([int(p[k], 0)])[0] for k in ("tau", "gscal", "chal"))
    lib = prover.load_host_library()
    buf = np.frombuffer(proof, dtype=np.uint8).copy()
    assert lib.plonk_proof_roundtrip(buf.ctypes.data_as(C.c_void_p), C.c_uint64(len(buf))) == 0
    assert prover.verify(tau, gs, proof, chal) is True
    # any change to an evaluation, a commitment, an opening, the challenge or the trapdoor must be rejected
    bad = bytearray(proof); bad[0] ^= 1
    assert prover.verify(tau, gs, bytes(bad), chal) is False
    bad = bytearray(proof); bad[9 * 32 + 9 * 33 + 33 + 5] ^= 1          # random_v of the first opening
    assert prover.verify(tau, gs, bytes(bad), chal) is False
    assert prover.verify(tau, gs, proof, synth.to_mont([int(p["chal"], 0) + 1])[0]) is False
    assert prover.verify(synth.to_mont([int(p["tau"], 0) + 1])[0], gs, proof, chal) is False
    # swapping two commitments keeps every point valid but breaks the openings
    bad = bytearray(proof); o = 9 * 32
    bad[o:o + 33], bad[o + 33:o + 66] = proof[o + 33:o + 66], proof[o:o + 33]
    assert prover.verify(tau, gs, bytes(bad), chal) is False


def test_host_code_under_sanitizers(tmp_path):
    """plonk_amd/host (setup.cpp, verifier.cpp, transcript.cpp) compiled with -fsanitize=address,undefined and driven through the rng,
    the point sampling, the sponge, one pairing check, proof decoding (good, corrupted, truncated) and a transcript challenge."""
    import json
    import subprocess
    exe = str(tmp_path / "host_sanitize")
    host = os.path.join(ROOT, "plonk_amd", "host")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(host, "setup.cpp"), os.path.join(host, "verifier.cpp"),
                           os.path.join(host, "transcript.cpp")])
    proof_hex = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))[0]["proof_hex"]
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, proof_hex], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.split("\n")
    assert lines[0] == "10719222850664546238" and "pairing 1" in lines and "roundtrip 1" in lines
    assert "rejected: invalid field element encoding" in lines and "no throw" not in lines
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_circuit_builder_of_the_host_library():
    """CircuitBuilder / Circuit::from_builder / WitnessTable::flatten / compute_sigma_mapping (src/circuit.rs:80-112,169-276) restated in
    C++ (plonk_amd/host/circuit.cpp) against the numpy tables of plonk_amd/synth.py and the reference's known answers for its example."""
    from plonk_amd import prover
    keys = ("a", "b", "c", "q_add_vals", "q_mul_vals", "witness_flat", "sigma")
    for variables, gates in (synth.example_circuit(), synth.synthetic_circuit(1, seed=2), synth.synthetic_circuit(5, seed=3), synth.synthetic_circuit(11, seed=4)):
        want, got = synth.circuit_tables(variables, gates), prover.circuit_tables(variables, gates)
        assert want["n"] == got["n"]
        for k in keys:
            assert np.array_equal(want[k], got[k]), k
    # five gates in a domain of eight rows: zero padding, identity sigma on the unused rows
    variables, gates = synth.synthetic_circuit(3, seed=9)
    variables, gates = variables[:7], gates[:5]
    want, got = synth.circuit_tables(variables, gates), prover.circuit_tables(variables, gates)
    for k in keys:
        assert np.array_equal(want[k], got[k]), k
    assert list(got["sigma"][15:]) == list(range(15, 24))
    ex = prover.circuit_tables(*synth.example_circuit())
    assert list(ex["sigma"]) == [10, 1, 3, 2, 4, 6, 5, 7, 9, 8, 0, 11]
    assert synth.from_mont(ex["witness_flat"]) == [3, 4, 7, 7, 2, 14, 14, 5, 19, 19, 3, 57]
    # a gate whose stated output value is wrong is refused (add_gate computes the value itself)
    from plonk_amd.hip import PlonkHipError
    bad = list(synth.example_circuit()[0]); bad[4] += 1
    with pytest.raises(PlonkHipError):
        prover.circuit_tables(bad, synth.example_circuit()[1])
