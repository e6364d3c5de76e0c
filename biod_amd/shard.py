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


def depth_histogram(coverage: np.ndarray) -> np.ndarray:
    """hist[d] = number of positions with coverage d (d clipped to DEPTH_BINS-1), as the device computes it."""
    return np.bincount(np.minimum(coverage.astype(np.int64), DEPTH_BINS - 1), minlength=DEPTH_BINS).astype(np.int64)


def allreduce_histogram(hist_tensor):
    """Sum a depth histogram (torch int64 tensor, on the rank's device) over all ranks in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist_tensor, op=dist.ReduceOp.SUM)
    return hist_tensor
