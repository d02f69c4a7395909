"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Every utterance is an independent decode (the reference runs one ThreadPool task per item,
ctcdecode/src/ctc_beam_search_decoder.cpp:259-275), so the batch is cut into contiguous blocks of ceil(B / world)
utterances, each rank decodes its block with no communication, and the four result tensors are gathered once to the
destination rank -- a point-to-point pattern: every peer has its own xGMI link to the root, no ring is involved.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, world, rank):
    """[lo, hi) of the utterances rank ``rank`` decodes; blocks of ceil(batch / world), the last ones may be short/empty."""
    per = (batch + world - 1) // world
    lo = min(rank * per, batch)
    return lo, min(lo + per, batch)


def gather_results(results, batch, dst=0, group=None):
    """``results`` = (output[b,K,T], scores[b,K], timesteps[b,K,T], out_lens[b,K]) of this rank's shard.
    Returns the four full-batch tensors on rank ``dst`` (None elsewhere).  Shards are zero-padded to equal size for the
    collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = (batch + world - 1) // world
    out = []
    for t in results:
        if t.shape[0] < per:
            pad = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], 0)
        t = t.contiguous()
        if rank == dst:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.gather(t, parts, dst=dst, group=group)
            out.append(torch.cat(parts, 0)[:batch])
        else:
            dist.gather(t, None, dst=dst, group=group)
    return tuple(out) if rank == dst else None


def decode_sharded(decode_fn, probs, seq_lens=None, dst=0, group=None):
    """Decode a full batch that every rank holds (or can index): rank r decodes ``probs[lo:hi]`` with
    ``decode_fn(probs_shard, seq_lens_shard) -> (output, scores, timesteps, out_lens)`` and the results are gathered to
    ``dst``.  With ``decode_fn = CTCBeamDecoder.decode_device`` the tensors stay in HBM and travel over RCCL/xGMI."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = probs.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    res = decode_fn(probs[lo:hi], None if seq_lens is None else seq_lens[lo:hi])
    return gather_results(res, B, dst=dst, group=group)
