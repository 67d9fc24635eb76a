"""Multi-GPU sharding of the block-compression path.

Every <=64 KiB block is independent (fresh hash table per fragment), so the path shards by
contiguous block ranges with NO data-path collective.  The only exchange is the final gather of
per-block compressed lengths (uint32 per block) so that every rank knows the global layout of the
packed stream: one all_gather over RCCL/xGMI (backend "nccl" on ROCm), or gloo on CPU in tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_blocks_total, rank, world):
    """Contiguous, balanced block range [begin, end) of `rank` (first ranks get the remainder)."""
    base, rem = divmod(n_blocks_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def check_lengths(lens, slot_stride=None):
    """Raises if a per-block length is one of the kernels' markers (0xffffffff: block longer than 64 KiB;
    0xfffffffe: poisoned result), which read as negative int32, or exceeds the slot stride.
    One device synchronisation: call it where the lengths are consumed, not inside a timed loop."""
    bad = lens < 0
    if slot_stride is not None:
        bad = bad | (lens > slot_stride)
    if bool(bad.any().item()):
        first = int(torch.nonzero(bad)[0].item())
        raise ValueError("block %d came back with the error marker / length %d" % (first, int(lens[first].item())))


def gather_lengths(local_lens, n_blocks_total=None, group=None, out=None, offsets=None, check=False):
    """all_gather of per-block lengths.  local_lens: int32 tensor [blocks of this rank] (CUDA for
    nccl, CPU for gloo).  Returns (all_lens int32[n_total], offsets int64[n_total+1]) where
    offsets is the exclusive prefix sum = position of every block in the packed global stream.
    Ranks may hold different block counts (shard_range); shorter ones are padded for the collective.
    `out` / `offsets`: optional preallocated results (a steady-state caller passes them so that a
    step is one collective and one scan, with no allocation).  `check`: raise on error markers in the
    gathered lengths (check_lengths: synchronises) instead of returning offsets computed from them."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        all_lens = local_lens
        if out is not None:
            out.copy_(local_lens)
            all_lens = out
    else:
        world = dist.get_world_size(group)
        if n_blocks_total is None:
            n_blocks_total = int(local_lens.numel()) * world
        if n_blocks_total == int(local_lens.numel()) * world:
            # equal shards (the benchmark's weak-scaling layout): gather straight into the result
            all_lens = out if out is not None else torch.empty(n_blocks_total, dtype=local_lens.dtype,
                                                               device=local_lens.device)
            dist.all_gather_into_tensor(all_lens, local_lens, group=group)
        else:
            per = -(-n_blocks_total // world)
            pad = torch.zeros(per, dtype=local_lens.dtype, device=local_lens.device)
            pad[:local_lens.numel()] = local_lens
            buf = torch.empty(per * world, dtype=local_lens.dtype, device=local_lens.device)
            dist.all_gather_into_tensor(buf, pad, group=group)
            pieces = []
            for r in range(world):
                b, e = shard_range(n_blocks_total, r, world)
                pieces.append(buf[r * per:r * per + (e - b)])
            all_lens = torch.cat(pieces)
            if out is not None:
                out.copy_(all_lens)
                all_lens = out
    if check:
        check_lengths(all_lens)
    if offsets is None:
        offsets = torch.zeros(all_lens.numel() + 1, dtype=torch.int64, device=all_lens.device)
    else:
        offsets[:1].zero_()
    torch.cumsum(all_lens, dim=0, dtype=torch.int64, out=offsets[1:])
    return all_lens, offsets
