#!/usr/bin/env python3
"""Checks that libplonk_hip and RCCL (torch.distributed backend "nccl") work in one process on this box: the library is used, a
one-rank process group is created, the two collectives of the multi-GPU paths (all_gather_into_tensor for MSM partial sums,
all_to_all_single for the four-step NTT) run on device tensors that the library wrote, and the library is used again.

    MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/rccl_probe.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from plonk_amd import synth, dist_ntt
from plonk_amd.hip import Context

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
ctx = Context(0)
tau, gs = synth.seeded_scalars(2, 7)
ctx.srs_generate(tau, gs, 1 << 12)
sc = synth.seeded_scalars(1 << 12, 3)
d = ctx.upload(sc)
part = ctx.msm_partial_dev(d, 1 << 12)
send = torch.from_numpy(part.view(np.int64)).cuda()
recv = torch.empty_like(send)
dist.all_gather_into_tensor(recv, send)
assert np.array_equal(recv.cpu().numpy().view(np.uint64), part)
ops = dist_ntt.HipOps(ctx, 0, backend="nccl")
buf, tmp = ops.alloc(1 << 12), ops.alloc(1 << 12)
ops.load(buf, sc)
ops.ntt(buf, 12, False, None)
ops.exchange(buf, tmp)                      # all_to_all_single with one rank: a device-to-device copy through RCCL
want = ctx.ntt(sc)
assert np.array_equal(ops.read(tmp), want)
again = ctx.msm_partial_dev(d, 1 << 12)   # the projective representative may differ, the point may not
assert np.array_equal(ctx.sum_partials(again[None])[0], ctx.sum_partials(part[None])[0])
dist.barrier()
dist.destroy_process_group()
print("rccl probe ok")
