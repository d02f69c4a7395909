"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Every utterance is an independent decode (the reference runs one ThreadPool task per item,
ctcdecode/src/ctc_beam_search_decoder.cpp:259-275), so the batch is cut into contiguous blocks of ceil(B / world)
utterances, each rank decodes its block with no communication, and the four result tensors are gathered once to the
destination rank -- a point-to-point pattern: every peer has its own xGMI link to the root, no ring is involved.
"""
import torch
import torch.distributed as dist


def shard_size(batch, world, min_shard=0):
    """Utterances per rank: ceil(batch / world), but not fewer than ``min_shard``.

    The partition rule: one utterance occupies one workgroup = one CU, and the time axis cannot be
    split (a true recurrence), so a shard smaller than the CU count leaves CUs idle without finishing any sooner -- 128
    utterances take 7.7 ms on half of an MI355X's 256 CUs, 256 take the same wall time on all of them.  ``min_shard`` = the
    CU count fills GPUs before it adds ranks: BASELINE's configs[4] (1024 utterances) then runs on 4 ranks x 256 instead of
    8 x 128 in the same time, and the other ranks get empty shards (free for the next batch)."""
    per = (batch + world - 1) // world
    return max(per, int(min_shard)) if min_shard else per


def shard_bounds(batch, world, rank, min_shard=0):
    """[lo, hi) of the utterances rank ``rank`` decodes; blocks of ``shard_size``, the last ones may be short/empty."""
    per = shard_size(batch, world, min_shard)
    lo = min(rank * per, batch)
    return lo, min(lo + per, batch)


def gather_results(results, batch, dst=0, group=None, min_shard=0):
    """``results`` = (output[b,K,T], scores[b,K], timesteps[b,K,T], out_lens[b,K]) of this rank's shard.
    Returns the four full-batch tensors on rank ``dst`` (None elsewhere).  Shards are zero-padded to equal size for the
    collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = shard_size(batch, world, min_shard)
    out = []
    host_only = dist.get_backend(group) == "gloo"  # gloo gathers host tensors only
    for t in results:
        if host_only and t.is_cuda:
            t = t.cpu()
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


def decode_sharded(decode_fn, probs, seq_lens=None, dst=0, group=None, min_shard=0):
    """Decode a full batch that every rank holds (or can index): rank r decodes ``probs[lo:hi]`` with
    ``decode_fn(probs_shard, seq_lens_shard) -> (output, scores, timesteps, out_lens)`` and the results are gathered to
    ``dst``.  With ``decode_fn = CTCBeamDecoder.decode_device`` the tensors stay in HBM and travel over RCCL/xGMI."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = probs.shape[0]
    lo, hi = shard_bounds(B, world, rank, min_shard)  # (min_shard: fill GPUs before adding ranks -- shard_size)
    res = decode_fn(probs[lo:hi], None if seq_lens is None else seq_lens[lo:hi])
    return gather_results(res, B, dst=dst, group=group, min_shard=min_shard)


class ResultGatherer(object):
    """Pipelined gather of per-rank result shards to ``dst``: ``submit(results)`` launches the four gathers
    asynchronously (they run on RCCL's own streams and only wait for the decode that produced ``results``), so the next
    batch can be decoded while the previous one's results travel over xGMI; ``wait()`` drains everything submitted.
    Receive buffers on ``dst`` are allocated once and reused (collectives of one process group execute in order)."""

    def __init__(self, shard_shapes_dtypes, device, dst=0, group=None, depth=2):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.depth = max(1, depth)
        self._inflight = []
        self._recv = None
        self._host_only = dist.get_backend(group) == "gloo"  # gloo gathers host tensors only (single-device dry runs)
        if self._host_only:
            device = torch.device("cpu")
        if self.rank == dst:
            self._recv = [[[torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world)] for shape, dtype in shard_shapes_dtypes]
                          for _ in range(self.depth)]
        self._n = 0

    def submit(self, results):
        slot = self._n % self.depth
        while len(self._inflight) >= self.depth:  # a receive slot is reused only after its previous gather completed
            works, _ = self._inflight.pop(0)
            for wk in works:
                wk.wait()
        works = []
        for i, t in enumerate(results):
            if self._host_only and t.is_cuda:
                t = t.cpu()
            works.append(dist.gather(t.contiguous(), self._recv[slot][i] if self.rank == self.dst else None, dst=self.dst,
                                     group=self.group, async_op=True))
        self._inflight.append((works, results))  # keep the source tensors alive until the collective has read them
        self._n += 1
        return slot

    def wait(self):
        while self._inflight:
            works, _ = self._inflight.pop(0)
            for wk in works:
                wk.wait()

    def received(self, slot):
        """On ``dst``: the per-rank parts of the batch submitted into ``slot`` (valid after its gather completed)."""
        return self._recv[slot] if self._recv is not None else None


class CompactGatherer(object):
    """Gather of COMPACT results (include/ctcdecode_amd.h "Compact result delivery"): every rank ships, per beam entry, only
    the labels it does not share with its neighbour in the trie -- an order of magnitude fewer bytes than the padded
    [B, K, T] pair -- and ``dst`` rebuilds the padded tensors of the whole batch in its HBM with one expansion kernel.

    Per batch: the ranks tell each other how many labels they hold -- one all_gather of a single host word over a gloo
    control group, so that every rank pads its label buffer to the same length and ``dst`` knows where each rank's labels
    end WITHOUT reading anything back from the device (no ``.item()`` / stream synchronisation in the gather path) --
    then five gathers to ``dst`` over the data group (RCCL; point to point underneath: every peer has its own xGMI link to
    the root).  ``submit`` launches them asynchronously; ``wait`` drains and, on ``dst``, leaves the expanded tensors of
    the last batch in ``self.last`` = (output, scores, timesteps, out_lens)."""

    def __init__(self, decoder, T, dst=0, group=None, depth=2, stream=None, ctl_group=None):
        self.dec, self.T, self.dst, self.group = decoder, T, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.depth = max(1, depth)
        self._host_only = dist.get_backend(group) == "gloo"
        # control plane: host integers travel over gloo.  The control group spans exactly the ranks of the data group (ADVICE
        # r3: a world-wide control group under a sub-group gave all_gather a list of the wrong length, and ranks outside the
        # sub-group never made the call).  NOTE: creating it is a collective -- torch.distributed.new_group must be called by
        # EVERY rank of the default group in the same order, also by ranks outside ``group``; callers that decode on a
        # sub-group either pass ``ctl_group`` (made up front on all ranks) or construct their gatherers on all ranks.
        if self._host_only:
            self.ctl = group
        elif ctl_group is not None:
            self.ctl = ctl_group
        elif group is None or group is dist.group.WORLD:
            self.ctl = dist.new_group(backend="gloo")
        else:
            self.ctl = dist.new_group(ranks=dist.get_process_group_ranks(group), backend="gloo")
        assert dist.get_world_size(self.ctl) == self.world, "control group and data group must span the same ranks"
        self._inflight = []
        self.last = None
        # Optional side stream (device backends): the collectives, and on ``dst`` the expansion kernel, are issued there
        # behind the event of the batch they belong to, so they neither wait for nor delay the NEXT batch's decode kernel,
        # which the caller has already queued on its own stream.
        self.stream = None if self._host_only else stream

    def _on_stream(self):
        import contextlib

        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _xfer(self, t):
        return t.cpu() if self._host_only and t.is_cuda else t

    def submit(self, compact, ready=None):
        """``ready``: event recorded behind the batch's decode (``decode_compact_async``); with a side stream the gather
        waits for that event instead of for everything queued on the caller's stream."""
        with self._on_stream():
            if self.stream is not None:
                if ready is not None:
                    self.stream.wait_event(ready)
                else:
                    self.stream.wait_stream(torch.cuda.current_stream(compact[0].device))
                for t in compact:
                    t.record_stream(self.stream)
            self._submit(compact)

    def _submit(self, compact):
        hdr, ent, labels, scores, lens = compact
        while len(self._inflight) >= self.depth:
            self._finish_one(self._inflight.pop(0))
        dev = hdr.device
        # the label buffers differ in length: every rank learns every count (host words over the control group), pads to
        # the longest, gathers
        mine = torch.tensor([labels.numel()], dtype=torch.int64)
        counts = [torch.zeros((1,), dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(counts, mine, group=self.ctl)
        ns = [int(c[0]) for c in counts]
        nmax = max(max(ns), 1)
        lab = labels
        if lab.numel() < nmax:
            lab = torch.cat([lab, torch.zeros((nmax - lab.numel(),), dtype=lab.dtype, device=lab.device)])
        send = [self._xfer(t).contiguous() for t in (hdr, ent, lab, scores, lens)]
        recv, works = None, []
        if self.rank == self.dst:
            recv = [[torch.empty_like(t) for _ in range(self.world)] for t in send]
        for i, t in enumerate(send):
            works.append(dist.gather(t, recv[i] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True))
        self._inflight.append((works, recv if self.rank == self.dst else send, dev, ns))  # (sources stay alive until the gather is done)

    def _finish(self, item):
        with self._on_stream():
            self._finish_one(item)

    def _finish_one(self, item):
        works, recv, dev, ns = item
        for wk in works:
            wk.wait()
        if self.rank != self.dst:
            return
        hdrs, ents, labs = [], [], []
        base = 0
        for r in range(self.world):
            h, e, lab = recv[0][r].to(dev), recv[1][r].to(dev).clone(), recv[2][r].to(dev)
            n = ns[r]  # (= the sum of the rank's per-item label counts h[:, 1]; known on the host since submit)
            e[:, :, 3] += base  # label indices (relative to the rank's own buffer) rebased onto the concatenated one
            hdrs.append(h); ents.append(e); labs.append(lab[:n])
            base += n
        hdr, ent = torch.cat(hdrs, 0), torch.cat(ents, 0)
        labels = torch.cat(labs, 0) if base else torch.zeros((1,), dtype=torch.int32, device=dev)
        out, ts = self.dec.expand_compact(hdr, ent, labels, self.T)
        self.last = (out, torch.cat([t.to(dev) for t in recv[3]], 0), ts, torch.cat([t.to(dev) for t in recv[4]], 0))

    def wait(self):
        while self._inflight:
            self._finish(self._inflight.pop(0))
        if self.stream is not None:  # what ``self.last`` holds was produced on the side stream
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)


def make_gatherer(fmt, B, K, T, V, device, dst=0, group=None, depth=2, decoder=None, stream=None, ctl_group=None):
    """The gatherer bench.py / a serving loop uses: ``fmt`` = "full" (the four padded tensors travel) or "compact"
    (the trie-compact form travels; needs the ``decoder`` to expand it on ``dst``)."""
    if fmt == "full":
        shapes = [((B, K, T), torch.int32), ((B, K), torch.float32), ((B, K, T), torch.int32), ((B, K), torch.int32)]
        return ResultGatherer(shapes, device, dst=dst, group=group, depth=depth)
    if fmt == "compact":
        return CompactGatherer(decoder, T, dst=dst, group=group, depth=depth, stream=stream, ctl_group=ctl_group)
    raise ValueError("unknown gather format %r" % (fmt,))
