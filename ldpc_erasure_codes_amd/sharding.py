"""Multi-GPU plumbing: one process per GPU, frames sharded with no data-path collective, one gather of the
per-frame status words at the end.

Frames are independent (one decoder call per frame in the reference, Matlab/ErasureCodes_NonBinaryLDPCSim.m:218;
the FPGA's per-frame while(1) loop, OpenCL/device/ldpc_erasure_decoder_perf_tests.cl:52), so sharding is a
partition of frame indices.  Everything here is backend-agnostic torch.distributed (RCCL on the GPUs, gloo
in the CPU tests).
"""
import numpy as np


def shard_frames(total, rank, world):
    """Contiguous block of frames for `rank`: (frame0, count).  The first total % world ranks get one more."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    frame0 = rank * base + min(rank, rem)
    return frame0, count


def shard_mixed(code_ids, rank, world):
    """Mixed stream (BASELINE cfg 5): bucket the frames by code id (kernels are specialised per code), then split
    every bucket evenly.  Returns {code_id: ascending array of global frame indices handled by this rank}."""
    code_ids = np.asarray(code_ids)
    out = {}
    for cid in np.unique(code_ids):
        idx = np.nonzero(code_ids == cid)[0]
        f0, cnt = shard_frames(idx.size, rank, world)
        out[int(cid)] = idx[f0:f0 + cnt]
    return out


def gather_status(words, counts=None):
    """Final gather of the status words.  words: int32 tensor [W, F_local] (sweeps / residual / status rows) on this
    rank's device.  Returns a list of per-rank tensors on every rank (ragged shards are padded for the collective
    and trimmed again).  With world size 1 it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [words]
    world = dist.get_world_size()
    dev = words.device
    if dist.get_backend() == "gloo":  # CPU rehearsals of the multi-GPU path: gloo gathers host tensors
        words = words.cpu()
    if counts is None:
        c = torch.tensor([words.shape[1]], dtype=torch.int64, device=words.device)
        allc = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(allc, c)
        counts = [int(x.item()) for x in allc]
    fmax = max(counts)
    padded = torch.zeros((words.shape[0], fmax), dtype=words.dtype, device=words.device)
    padded[:, :words.shape[1]] = words
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    return [gathered[r][:, :counts[r]].to(dev) for r in range(world)]


def max_over_ranks(seconds, device=None):
    """The job's step time is the slowest rank's."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
