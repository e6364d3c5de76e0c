"""Multi-GPU decomposition of the pileup path: shard by reference contig, reduce only histograms.

A pileup column at (ref_id, pos) depends only on reads of that reference (pileup.d:352,377-379), so
contigs are independent units: every rank decodes and piles up its own contigs with no data-path
collective, and the per-rank column arrays simply concatenate in contig order.  The single exchange
of the path is the whole-BAM depth histogram (a fold over column.coverage, pileup.d:252), summed
with one all-reduce -- RCCL over xGMI on GPUs (`backend="nccl"`), gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .synth import CONFIGS, make_cfg, scaled

DEPTH_BINS = L.BGP_DEPTH_BINS


def weak_cfg(name: str, reads_per_gpu: int, rank: int) -> L.SynthCfg:
    """Weak scaling: every rank gets a full-size shard made of its own contigs
    (global contig ids rank*n_ref .. rank*n_ref+n_ref-1)."""
    return scaled(name, reads_per_gpu, first_ref=rank * CONFIGS[name]["n_ref"])


def strong_cfg(name: str, total_reads: int, rank: int, world: int) -> L.SynthCfg:
    """Strong scaling (BASELINE.json config 4): the SAME BAM split by contig.  Records are pure
    functions of (global contig, index in contig), so rank r's stream is byte-identical to the slice
    of the whole stream that holds its contigs."""
    whole = scaled(name, total_reads)
    n_ref = whole.n_ref
    if n_ref % world:
        raise ValueError(f"{n_ref} contigs do not split evenly over {world} ranks")
    per_contig = (whole.n_reads + n_ref - 1) // n_ref
    mine = n_ref // world
    first = rank * mine
    n = min(per_contig * mine, max(whole.n_reads - per_contig * first, 0))
    return make_cfg(n_reads=n, n_ref=mine, ref_len=whole.ref_len, read_len=(whole.len_min, whole.len_max),
                    profile=whole.profile, seed=whole.seed, first_ref=first)


def plan_interval_shards(ref_id: np.ndarray, pos: np.ndarray, end_pos: np.ndarray, rec_off: np.ndarray, stream_len: int, world: int):
    """Equal-read-count shards INSIDE contigs, with the exact halo ("refinement for balance").

    Shard r owns the records [cut[r], cut[r+1]) -- cuts at multiples of n / world, moved forward to the first record
    of a position group so that reads starting at one position stay together -- and hands out the columns from the
    (ref, position) of its first read up to the (ref, position) of the next shard's first read.  The reads it needs are
    its own plus every earlier read of the same reference whose end_position exceeds the shard's first position: the
    first record whose running maximum of end_position does (the rule the windows of K4 and the batches use; BioD's own
    rule, pileup.d:944-985, keeps reads that start within twice a sampled read length and is not exact).  That set is a
    contiguous range of records, so a shard is a byte range of the stream plus a clip.

    Returns a list of dicts: byte_lo, byte_hi (the sub-stream), first_record (ordinal of its record 0), lo, hi
    ((ref_key, position) keys: keep columns with lo <= key < hi; ref_key = ref_id as unsigned, unmapped last)."""
    n = len(pos)
    key = (np.where(ref_id < 0, 0x7FFFFFFF, ref_id.astype(np.int64) + 1) << 32) | (pos.astype(np.int64) & 0xFFFFFFFF)
    covering = end_pos != pos
    cuts = [0]
    for r in range(1, world):
        c = min(max(r * n // world, cuts[-1]), n)
        while c < n and (not covering[c] or (c > 0 and key[c] == key[c - 1])):       # a shard starts with the first read of a position group, a covering one
            c += 1
        cuts.append(c)
    cuts.append(n)
    ends = np.append(rec_off[1:], stream_len).astype(np.int64) if n else np.zeros(0, np.int64)
    shards = []
    top = (0x7FFFFFFF + 1) << 32
    for r in range(world):
        a, b = cuts[r], cuts[r + 1]
        if a >= b:
            shards.append(dict(byte_lo=0, byte_hi=0, first_record=a, lo=top, hi=top))
            continue
        lo_key = int(key[a]) if r > 0 else 0
        hi_key = int(key[b]) if b < n else top
        i_lo = a
        if r > 0:
            same = np.nonzero(ref_id[:a] == ref_id[a])[0]                       # earlier records of the shard's first reference
            if len(same):
                f = int(same[0])
                run_max = np.maximum.accumulate(np.where(covering[f:a], end_pos[f:a], np.iinfo(np.int32).min).astype(np.int64))
                hit = np.nonzero(run_max > int(pos[a]))[0]
                if len(hit):
                    i_lo = f + int(hit[0])
        shards.append(dict(byte_lo=int(rec_off[i_lo]), byte_hi=int(ends[b - 1]), first_record=i_lo, lo=lo_key, hi=hi_key))
    return shards


def clip_columns(ref: np.ndarray, pos: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """Mask of the columns (ref, pos) a shard of plan_interval_shards hands out."""
    key = (np.where(ref < 0, 0x7FFFFFFF, ref.astype(np.int64) + 1) << 32) | (pos.astype(np.int64) & 0xFFFFFFFF)
    return (key >= lo) & (key < hi)


def depth_histogram(coverage: np.ndarray) -> np.ndarray:
    """hist[d] = number of positions with coverage d (d clipped to DEPTH_BINS-1), as the device computes it."""
    return np.bincount(np.minimum(coverage.astype(np.int64), DEPTH_BINS - 1), minlength=DEPTH_BINS).astype(np.int64)


def allreduce_histogram(hist_tensor):
    """Sum a depth histogram (torch int64 tensor, on the rank's device) over all ranks in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist_tensor, op=dist.ReduceOp.SUM)
    return hist_tensor


# ---------------------------------------------------------------------------------------------- a BAM FILE over N ranks
# BioD's pieces for this: ReferenceSequence.startVirtualOffset (reference.d:96-104), Index.getMinimumOffset
# (baifile.d:77-82) and bam[ref][beg .. end] (randomaccessmanager.d:222-244,366-462).  A rank that loads a REGION gets
# every read overlapping it, so its columns inside [beg, end) are exact without any halo arithmetic: cutting a file is
# choosing cut points, and the linear index says how many compressed bytes lie in front of each candidate.

LINEAR_SHIFT = 14      # the BAI linear index has one entry per 16 384 positions (bai/indexing.d, samtools spec)


def plan_file_shards(bai, ref_lengths, world: int):
    """Regions (ref_id, beg, end) for each of `world` ranks: the references in file order, cut at linear-index windows
    so that the COMPRESSED bytes of the ranks balance.  Returns a list of `world` lists; a reference without an entry in
    the index (no reads) is given to nobody.  Every position of every indexed reference lies in exactly one region."""
    cand = []                                           # (compressed offset, ref_id, position): possible region starts, file order
    for r, rlen in enumerate(ref_lengths):
        if r >= bai.n_ref or bai.shape(r)[2] == 0 or rlen <= 0:
            continue
        n_win = (int(rlen) + (1 << LINEAR_SHIFT) - 1) >> LINEAR_SHIFT
        last = -1
        for w in range(n_win):
            off = bai.min_offset(r, w << LINEAR_SHIFT) >> 16
            if w == 0 or off > last:                    # (windows without a read of their own repeat their neighbour's offset: no cut there)
                cand.append((max(off, last), r, w << LINEAR_SHIFT))
                last = max(off, last)
    shards = [[] for _ in range(world)]
    if not cand:
        return shards
    lo, hi = cand[0][0], cand[-1][0]
    span = max(hi - lo, 1)
    # the first candidate at or beyond k/world of the span starts rank k's share
    starts = [0]
    for k in range(1, world):
        target = lo + span * k // world
        j = starts[-1]
        while j < len(cand) and cand[j][0] < target:
            j += 1
        starts.append(j)
    starts.append(len(cand))
    for k in range(world):
        a, b = starts[k], starts[k + 1]
        j = a
        while j < b:
            r = cand[j][1]
            beg = cand[j][2]
            while j < b and cand[j][1] == r:
                j += 1
            end = cand[j][2] if j < len(cand) and cand[j][1] == r else int(ref_lengths[r])     # up to the next rank's first cut in this reference, or its end
            shards[k].append((r, int(beg), int(end)))
    return shards


def file_shard_windows(ctx, path: str, bai, regions, n_threads: int = 0, device_inflate: bool = False, **window_args):
    """The windows of one rank's regions: bam[ref][beg .. end] staged through the index, piled up, and the columns of
    [beg, end) handed out (makePileup's start_from / end_at, pileup.d:683-694).  device_inflate: the chunks' BGZF blocks are
    inflated by the GPU (bgp_load_bam_region_device) -- with a rank per GPU the host's cores are shared by all of them."""
    for ref_id, beg, end in regions:
        ctx.reset()
        if device_inflate:
            ctx.load_bam_region_device(path, bai, ref_id, beg, end)
        else:
            ctx.load_bam_region(path, bai, ref_id, beg, end, n_threads)
        ctx.flush()
        if ctx.num_records() == 0:
            continue
        ctx.pileup_run()
        yield from ctx.pileup_windows(L.BGP_MAKE_PILEUP, start_from=beg, end_at=end, **window_args)
