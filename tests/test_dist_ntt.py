"""Four-step NTT across ranks (plonk_amd/dist_ntt.py): layouts, twiddles and the exchange step.

CPU part: the oracle stands in for the local device work, so the rank logic is checked without a GPU - G = 2, 4, 8 logical
ranks in one process, and two real ranks over gloo.  GPU part (marked gpu): the same walks with libplonk_hip doing the
local work, and two processes on one GPU exchanging through gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pymodel as pm
from plonk_amd import dist_ntt, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """Local steps on numpy arrays through the CPU oracle and Python integers."""

    def __init__(self, orc):
        self.orc = orc

    def ntt(self, buf, log_m, inverse, coset):
        buf[:] = self.orc.ntt(buf, inverse=inverse, coset=coset)

    def scale_powers(self, buf, count, base, ratio):
        b, r = synth.from_mont(base)[0], synth.from_mont(ratio)[0]
        vals = synth.from_mont(buf[:count])
        out, cur = [], b
        for v in vals:
            out.append(v * cur % pm.R_MOD)
            cur = cur * r % pm.R_MOD
        buf[:count] = synth.to_mont(out)

    def cross(self, buf, log_g, cols, inverse):
        g = 1 << log_g
        w = dist_ntt.root_of_unity(log_g) if log_g else 1
        if inverse:
            w = pow(w, -1, pm.R_MOD)
        vals = synth.from_mont(buf[:g * cols])
        out = [0] * (g * cols)
        for c in range(cols):
            for k1 in range(g):
                out[k1 * cols + c] = sum(vals[j * cols + c] * pow(w, j * k1, pm.R_MOD) for j in range(g)) % pm.R_MOD
        buf[:g * cols] = synth.to_mont(out)


def walk_logical_ranks(make_ops, load, read, world, log_n, coset, x_full, want_full):
    """Forward then inverse over `world` logical ranks in one process; returns nothing, asserts bit equality."""
    sn = [dist_ntt.ShardedNtt(make_ops(r), r, world, log_n, coset) for r in range(world)]
    bufs = [load(r, x_full[dist_ntt.coeff_indices(r, world, log_n)]) for r in range(world)]
    for r in range(world):
        sn[r].forward_local(bufs[r])
    if world > 1:
        moved = dist_ntt.exchange_blocks([read(r, bufs[r]) for r in range(world)])
        bufs = [load(r, moved[r]) for r in range(world)]
        for r in range(world):
            sn[r].forward_finish(bufs[r])
    for r in range(world):
        assert np.array_equal(read(r, bufs[r]), want_full[dist_ntt.eval_indices(r, world, log_n)]), f"forward, rank {r}"
    if world > 1:
        for r in range(world):
            sn[r].inverse_local(bufs[r])
        moved = dist_ntt.exchange_blocks([read(r, bufs[r]) for r in range(world)])
        bufs = [load(r, moved[r]) for r in range(world)]
    for r in range(world):
        sn[r].inverse_finish(bufs[r])
        assert np.array_equal(read(r, bufs[r]), x_full[dist_ntt.coeff_indices(r, world, log_n)]), f"inverse, rank {r}"


def _inputs(orc, log_n, seed, coset):
    rng = pm.SplitMix64(seed)
    x = orc.fr_from_ints([rng.next_fr() for _ in range(1 << log_n)])
    g = orc.fr_from_ints([coset])[0] if coset is not None else None
    return x, orc.ntt(x, coset=g)


@pytest.mark.parametrize("world,log_n,coset", [(1, 5, None), (2, 6, None), (2, 7, 5), (4, 7, 5), (8, 8, None), (8, 9, 5)])
def test_logical_ranks_with_oracle_ops(orc, world, log_n, coset):
    x, want = _inputs(orc, log_n, 31 + log_n, coset)
    ops = OracleOps(orc)
    walk_logical_ranks(lambda r: ops, lambda r, a: a.copy(), lambda r, b: b, world, log_n, coset, x, want)


def test_index_maps_are_permutations():
    for world in (1, 2, 4, 8):
        for fn in (dist_ntt.coeff_indices, dist_ntt.eval_indices):
            seen = np.concatenate([fn(r, world, 8) for r in range(world)])
            assert sorted(seen.tolist()) == list(range(256))


GLOO_RANK = r'''
import os, sys
import numpy as np
ROOT, use_gpu = sys.argv[1], sys.argv[2] == "gpu"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from oracle import oracle as orc, pymodel as pm
from plonk_amd import dist_ntt
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
log_n, coset = (14, 5) if use_gpu else (7, 5)
rng = pm.SplitMix64(99)
x = orc.fr_from_ints([rng.next_fr() for _ in range(1 << log_n)])
want = orc.ntt(x, coset=orc.fr_from_ints([coset])[0])
mine = np.ascontiguousarray(x[dist_ntt.coeff_indices(rank, world, log_n)])
if use_gpu:
    from plonk_amd.hip import Context
    ctx = Context(0)
    ops = dist_ntt.HipOps(ctx, 0, backend="gloo")
    buf, tmp = ops.alloc(len(mine)), ops.alloc(len(mine))
    ops.load(buf, mine)
    read = ops.read
else:
    from test_dist_ntt import OracleOps
    class Ops(OracleOps):
        def exchange(self, send, recv):
            out = torch.empty(send.size, dtype=torch.int64)
            dist.all_to_all_single(out, torch.from_numpy(send.view(np.int64).reshape(-1).copy()))
            recv[:] = out.numpy().view(np.uint64).reshape(recv.shape)
    ops = Ops(orc)
    buf, tmp = mine.copy(), np.empty_like(mine)
    read = lambda b: b
sn = dist_ntt.ShardedNtt(ops, rank, world, log_n, coset)
ev = sn.forward(buf, tmp)
assert np.array_equal(read(ev), want[dist_ntt.eval_indices(rank, world, log_n)]), "forward differs from the oracle"
other = buf if ev is tmp else tmp
back = sn.inverse(ev, other)
assert np.array_equal(read(back), mine), "inverse does not return the input"
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _run_two_ranks(tmp_path, mode, port):
    script = tmp_path / "rank.py"
    script.write_text(GLOO_RANK)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_two_ranks_over_gloo_with_oracle_ops(orc, tmp_path):
    _run_two_ranks(tmp_path, "cpu", 29641)


@pytest.mark.gpu
@pytest.mark.parametrize("world,log_n,coset", [(1, 12, 5), (4, 10, 5), (2, 12, None), (2, 13, 5), (4, 14, 5), (8, 15, 5), (8, 16, None)])
def test_logical_ranks_on_the_gpu(ctx, orc, world, log_n, coset):
    x, want = _inputs(orc, log_n, 61 + log_n, coset)
    ops = dist_ntt.HipOps(ctx, 0, backend="gloo")

    def load(r, a):
        b = ops.alloc(len(a))
        ops.load(b, a)
        return b
    walk_logical_ranks(lambda r: ops, load, lambda r, b: ops.read(b), world, log_n, coset, x, want)


@pytest.mark.gpu
def test_two_processes_on_one_gpu_over_gloo(orc, tmp_path):
    _run_two_ranks(tmp_path, "gpu", 29642)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,coset", [(12, 5), (16, None), (20, 5)])
def test_library_transform_on_a_one_rank_communicator(orc, log_n, coset):
    """plk_ntt_sharded_dev as one call (local transform, twiddles, grouped ncclSend / ncclRecv on the library's communicator,
    cross-rank butterflies, all on the library's stream) with the one rank a one-GPU box offers: the rank sends its block to
    itself, so the RCCL point-to-point path runs; forward equals the oracle's transform, inverse returns the input bit for bit.
    Without a communicator the call is refused."""
    from plonk_amd.hip import Context, PlonkHipError
    c = Context(0)
    x, want = _inputs(orc, log_n, 300 + log_n, coset)
    ops = dist_ntt.HipOps(c, 0, backend="nccl", library_exchange=True, self_exchange=True)
    buf, tmp = ops.alloc(len(x)), ops.alloc(len(x))
    ops.load(buf, x)
    sn = dist_ntt.ShardedNtt(ops, 0, 1, log_n, coset)
    with pytest.raises(PlonkHipError):
        sn.forward(buf, tmp)
    c.comm_init(c.comm_unique_id(), 0, 1)
    ev = sn.forward(buf, tmp)
    assert ev is tmp                                   # an exchange took place
    assert np.array_equal(ops.read(ev), want)
    back = sn.inverse(ev, buf)
    assert back is buf and np.array_equal(ops.read(back), x)
    del buf, tmp
    c.close()
