"""Four-step NTT of 2^log_n field elements over G = 2^log_g GPUs with ONE all-to-all per transform.

The reference has no counterpart (its largest transform is a single-threaded ark-poly fft of 4n points, src/poly_utils.rs:27-28);
this is the multi-GPU form of the standalone coset NTT.  N = G * M.  Layouts (one all-to-all means input and output
are distributed differently, like the bit-reversed output of an in-place FFT):

  coefficients  cyclic:        rank g holds c[g + G m]                      at local index m          (m < M)
  evaluations   column blocks: rank r holds X[(r M/G + k2l) + M k1]         at local index k1 M/G + k2l (k1 < G, k2l < M/G)

forward (X[k] = sum_i c[i] h^i w_N^(i k), h the coset generator or 1), with i = g + G m and k = k2 + M k1:
  X[k2 + M k1] = sum_g w_G^(g k1) * [ h^g w_N^(g k2) * sum_m c_g[m] (h^G)^m w_M^(m k2) ]
  1. local (coset) NTT of M points, generator h^G          2. times h^g (w_N^g)^k2
  3. all-to-all: block r of k2 goes to rank r               4. G-point transforms across the received blocks
inverse: the same steps backwards with inverse roots; 1/G rides on step 2, 1/M on the local inverse NTT.

`ops` supplies the local work on opaque buffers: HipOps (libplonk_hip through the C ABI, buffers are torch tensors so that
torch.distributed can move them) or any object with the same methods (the CPU tests plug in the oracle).

On the GPU the whole transform is one library call, plk_ntt_sharded_dev (include/plonk_hip.h): the steps above and the exchange
(grouped ncclSend / ncclRecv on the library's communicator) are enqueued on the library's stream, so a C++ or Rust host runs it
without Python.  This module keeps the index maps, the step-by-step form for the CPU tests (oracle as the local transform, gloo as
the transport) and the ranks-on-one-GPU rehearsals, which move the blocks themselves around the library's two halves.
"""
from __future__ import annotations

import numpy as np

from . import synth

R_MOD = synth.R_MOD
ROOT_2_28 = pow(5, (R_MOD - 1) >> 28, R_MOD)      # ark-ff's TWO_ADIC_ROOT_OF_UNITY of BN254 Fr


def root_of_unity(log_n: int) -> int:
    """ark-poly Radix2EvaluationDomain::group_gen for size 2^log_n."""
    return pow(ROOT_2_28, 1 << (28 - log_n), R_MOD)


def coeff_indices(rank: int, world: int, log_n: int) -> np.ndarray:
    """Global coefficient index of every local slot of `rank` (cyclic layout)."""
    m = (1 << log_n) // world
    return rank + world * np.arange(m, dtype=np.int64)


def eval_indices(rank: int, world: int, log_n: int) -> np.ndarray:
    """Global evaluation index (exponent of w_N) of every local slot of `rank` (column-block layout)."""
    m = (1 << log_n) // world
    blk = m // world
    k1 = np.arange(world, dtype=np.int64)[:, None]
    k2 = rank * blk + np.arange(blk, dtype=np.int64)[None, :]
    return (k2 + m * k1).reshape(-1)


class ShardedNtt:
    def __init__(self, ops, rank: int, world: int, log_n: int, coset: int | None = None):
        log_g = world.bit_length() - 1
        if 1 << log_g != world or world > 8:
            raise ValueError("world size must be 1, 2, 4 or 8")
        if log_n < 2 * log_g + 1 or log_n - log_g > 27:
            raise ValueError("transform too small or too large for this many ranks")
        self.ops, self.rank, self.world, self.log_n, self.log_g = ops, rank, world, log_n, log_g
        self.log_m = log_n - log_g
        self.m = 1 << self.log_m
        h = 1 if coset is None else coset % R_MOD
        w_n = root_of_unity(log_n)
        g = rank
        mont = lambda v: synth.to_mont([v])[0]
        self.local_coset = None if coset is None else mont(pow(h, world, R_MOD))
        self.fwd_base, self.fwd_ratio = mont(pow(h, g, R_MOD)), mont(pow(w_n, g, R_MOD))
        g_inv = pow(world, -1, R_MOD)
        self.inv_base = mont(g_inv * pow(h, -g, R_MOD) % R_MOD)
        self.inv_ratio = mont(pow(w_n, -g, R_MOD))
        self.trivial_twiddle = coset is None and g == 0
        self.coset_mont = None if coset is None else mont(h)
        self.in_library = hasattr(ops, "sharded_half")          # the library computes the factors above itself

    # The two halves around the all-to-all are separate methods so that a test can walk several logical ranks through them
    # on one device and move the blocks itself.
    def forward_local(self, buf):
        if self.in_library:
            return self.ops.sharded_half(buf, self.log_n, False, self.coset_mont, 1, self.rank, self.world)
        self.ops.ntt(buf, self.log_m, False, self.local_coset)
        if self.world > 1 and not self.trivial_twiddle:
            self.ops.scale_powers(buf, self.m, self.fwd_base, self.fwd_ratio)

    def forward_finish(self, tmp):
        if self.in_library:
            return self.ops.sharded_half(tmp, self.log_n, False, self.coset_mont, 2, self.rank, self.world)
        self.ops.cross(tmp, self.log_g, self.m // self.world, False)

    def inverse_local(self, buf):
        if self.in_library:
            return self.ops.sharded_half(buf, self.log_n, True, self.coset_mont, 1, self.rank, self.world)
        self.ops.cross(buf, self.log_g, self.m // self.world, True)

    def inverse_finish(self, tmp):
        if self.in_library:
            return self.ops.sharded_half(tmp, self.log_n, True, self.coset_mont, 2, self.rank, self.world)
        if self.world > 1:
            self.ops.scale_powers(tmp, self.m, self.inv_base, self.inv_ratio)
        self.ops.ntt(tmp, self.log_m, True, self.local_coset)

    def forward(self, buf, tmp):
        """buf: this rank's M coefficients (cyclic layout), destroyed.  Returns the buffer that holds the evaluations."""
        if getattr(self.ops, "library_exchange", False):
            return tmp if self.ops.sharded_whole(buf, tmp, self.log_n, False, self.coset_mont) else buf
        self.forward_local(buf)
        if self.world == 1:
            return buf
        self.ops.exchange(buf, tmp)
        self.forward_finish(tmp)
        return tmp

    def inverse(self, buf, tmp):
        """buf: this rank's M evaluations (column-block layout), destroyed.  Returns the buffer that holds the coefficients."""
        if getattr(self.ops, "library_exchange", False):
            return tmp if self.ops.sharded_whole(buf, tmp, self.log_n, True, self.coset_mont) else buf
        if self.world == 1:
            self.inverse_finish(buf)
            return buf
        self.inverse_local(buf)
        self.ops.exchange(buf, tmp)
        self.inverse_finish(tmp)
        return tmp


def exchange_blocks(bufs):
    """What the all-to-all does, for logical ranks held in one process: block r of rank g's buffer becomes block g of rank r's."""
    world = len(bufs)
    blk = bufs[0].shape[0] // world
    return [np.concatenate([bufs[g][r * blk:(r + 1) * blk] for g in range(world)]) for r in range(world)]


class HipOps:
    """Local steps on the GPU through the C ABI; buffers are torch int64 tensors of shape (count, 4) on the context's device.

    library_exchange = True (the context has a communicator, plk_comm_init): a transform is ONE call of plk_ntt_sharded_dev, the
    exchange runs as grouped ncclSend / ncclRecv on the library's stream.  Otherwise the halves come from the library and the
    blocks move through `exchange`: all_to_all_single over the default process group, device to device with backend "nccl", through
    host memory with "gloo" (several ranks rehearsing on one GPU)."""

    def __init__(self, ctx, device_index: int, backend: str = "nccl", library_exchange: bool = False, self_exchange: bool = False):
        import torch
        self.torch, self.ctx, self.backend = torch, ctx, backend
        self.device = torch.device("cuda", device_index)
        self.library_exchange = library_exchange
        self.phase = 3 if self_exchange else 0               # 3: a single rank still sends to itself (exercises the communicator on one GPU)

    def sharded_half(self, buf, log_n, inverse, coset, half, rank, world):
        self.ctx.ntt_sharded_dev(buf.data_ptr(), 0, log_n, inverse=inverse, coset=coset, phase=half, rank=rank, world=world)

    def sharded_whole(self, buf, tmp, log_n, inverse, coset) -> bool:
        return self.ctx.ntt_sharded_dev(buf.data_ptr(), tmp.data_ptr(), log_n, inverse=inverse, coset=coset, phase=self.phase)

    def alloc(self, count: int):
        return self.torch.empty((count, 4), dtype=self.torch.int64, device=self.device)

    def load(self, buf, arr: np.ndarray):
        buf.copy_(self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)))

    def read(self, buf) -> np.ndarray:
        self.ctx.sync()
        return buf.cpu().numpy().view(np.uint64)

    def ntt(self, buf, log_m, inverse, coset):
        self.ctx.ntt_dev_enqueue(buf.data_ptr(), log_m, inverse=inverse, coset=coset)

    def scale_powers(self, buf, count, base, ratio):
        self.ctx.scale_powers_dev(buf.data_ptr(), count, base, ratio)

    def cross(self, buf, log_g, cols, inverse):
        self.ctx.ntt_cross_dev(buf.data_ptr(), log_g, cols, inverse=inverse)

    def exchange(self, send, recv):
        import torch.distributed as dist
        self.ctx.sync()                      # the library's stream is not torch's
        if self.backend == "nccl":
            dist.all_to_all_single(recv.view(-1), send.view(-1))
            self.torch.cuda.synchronize(self.device)
        else:
            host_out = self.torch.empty(send.numel(), dtype=self.torch.int64)
            dist.all_to_all_single(host_out, send.view(-1).cpu())
            recv.view(-1).copy_(host_out)
            self.torch.cuda.synchronize(self.device)
