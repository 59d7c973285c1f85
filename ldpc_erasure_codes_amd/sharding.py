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


# ---------------------------------------------------------------------------------------------------------
# Mixed stream (BASELINE cfg 5): frames of several codes interleaved in one stream, sharded over the ranks
# ---------------------------------------------------------------------------------------------------------
def mixed_stream_ids(total, pattern=(2, 1)):
    """Code id of every frame of the stream: code B (4000,2000) and code A (2040,1530) interleaved 1:1
    (SURVEY.md 8d cfg 5); a ragged tail keeps the pattern."""
    reps = (total + len(pattern) - 1) // len(pattern)
    return np.tile(np.asarray(pattern, dtype=np.int32), reps)[:total]


def gather_rows(local, counts=None):
    """All-gather of ragged shards along dim 0.  local: tensor [F_local, ...] on this rank's device.  Every rank gets
    the list of all ranks' shards (rank order).  Shards are padded to the longest for the collective -- after
    shard_frames they differ by at most one row -- and trimmed again.  One collective: RCCL all_gather_into_tensor on
    the GPUs, gloo all_gather of host tensors in the CPU rehearsal."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size()
    dev = local.device
    gloo = dist.get_backend() == "gloo"
    if gloo:
        local = local.cpu()
    if counts is None:
        c = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        allc = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(allc, c)
        counts = [int(x.item()) for x in allc]
    fmax = max(counts)
    if fmax == 0:
        return [local[:0].to(dev) for _ in range(world)]
    padded = local
    if local.shape[0] != fmax:
        padded = torch.zeros((fmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    padded = padded.contiguous()
    if gloo:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
    else:
        flat = torch.empty((world,) + tuple(padded.shape), dtype=padded.dtype, device=padded.device)
        dist.all_gather_into_tensor(flat, padded)
        parts = [flat[r] for r in range(world)]
    return [parts[r][:counts[r]].to(dev) for r in range(world)]


def decode_mixed_shard(code_ids, rank, world, make_inputs, decode):
    """This rank's share of a mixed stream: bucket by code, take the rank's block of every bucket, decode bucket by bucket.
      make_inputs(cid, gidx) -> (sym, erased)         inputs of the frames with global stream indices gidx
      decode(cid, sym, erased) -> (out, sweeps, residual, status)
    Returns {cid: {"gidx", "out", "words" (int32 [3, F_local]: sweeps / residual / status)}}.  No collective."""
    import torch
    res = {}
    for cid, gidx in shard_mixed(code_ids, rank, world).items():
        sym, era = make_inputs(cid, gidx)
        out, sw, rs, st = decode(cid, sym, era)
        as_t = (lambda x: x) if torch.is_tensor(sw) else (lambda x: torch.from_numpy(np.ascontiguousarray(x)))
        words = torch.stack([as_t(sw).to(torch.int32), as_t(rs).to(torch.int32), as_t(st).to(torch.int32)])
        res[cid] = {"gidx": gidx, "out": as_t(out), "words": words}
    return res


def gather_mixed(code_ids, shard, world, what="status"):
    """The final gather of a mixed-stream job: per code, the status words -- and with what="outputs" the
    decoded frames too -- of all ranks, concatenated in ascending global frame order (the buckets are split into
    contiguous blocks in rank order, so rank order IS frame order).  Returns {cid: {"gidx", "words", "out" or None}}."""
    import torch
    code_ids = np.asarray(code_ids)
    res = {}
    for cid in sorted(int(c) for c in np.unique(code_ids)):
        idx = np.nonzero(code_ids == cid)[0]
        counts = [shard_frames(idx.size, r, world)[1] for r in range(world)]
        mine = shard[cid]
        words = torch.cat(gather_status(mine["words"], counts), dim=1)
        out = None
        if what == "outputs":
            out = torch.cat(gather_rows(mine["out"], counts), dim=0)
        res[cid] = {"gidx": idx, "words": words, "out": out}
    return res
