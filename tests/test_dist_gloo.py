"""The N>1 path on CPU: two gloo ranks shard a synthetic BAM by contig, pile up their shard (oracle
stands in for the device here -- this test is about the decomposition and the reduce), all-reduce the
depth histogram and must reproduce the single-process result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from biod_amd.shard import allreduce_histogram, clip_columns, depth_histogram, plan_interval_shards, strong_cfg, weak_cfg
from biod_amd.synth import host_stream, scaled


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dense_cov(st, skip_zero=False):
    return O.pileup(st, skip_zero=skip_zero)


def _worker(rank, world, port, total_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = strong_cfg("cfg3", total_reads, rank, world)
    st = host_stream(cfg)
    p = O.pileup(st, skip_zero=False)
    hist = torch.from_numpy(depth_histogram(p.cov))
    allreduce_histogram(hist)
    tot = torch.tensor([p.n_columns, p.sum_coverage, int((p.cov > 0).sum())], dtype=torch.int64)
    dist.all_reduce(tot)
    t = torch.tensor([0.01 * (rank + 1)], dtype=torch.float64)      # the bench's max-over-ranks timing idiom
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.concatenate([hist.numpy(), tot.numpy(), [int(round(t.item() * 100))], [len(st)]]))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_shards_concatenate_to_whole_stream():
    whole = host_stream(scaled("cfg3", 40_000))
    parts = b"".join(host_stream(strong_cfg("cfg3", 40_000, r, 4)) for r in range(4))
    assert parts == whole
    a = host_stream(weak_cfg("cfg3", 5_000, 0)); b = host_stream(weak_cfg("cfg3", 5_000, 1))
    assert a != b and len(O.frame(a)) == len(O.frame(b)) == 5_000


@pytest.mark.timeout(300)
def test_two_rank_histogram_reduce_matches_single_process(tmp_path):
    total = 24_000
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0[:-1], r1[:-1])
    whole = host_stream(scaled("cfg3", total))
    p = O.pileup(whole, skip_zero=False)
    assert np.array_equal(r0[:65536], depth_histogram(p.cov))
    assert list(r0[65536:65539]) == [p.n_columns, p.sum_coverage, int((p.cov > 0).sum())]
    assert r0[65539] == 2                                    # max over ranks of the timings
    assert r0[-1] + r1[-1] == len(whole)


def _interval_worker(rank, world, port, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = host_stream(scaled("cfg3", n_reads))                 # (every rank can make the stream: records are functions of their index)
    soa = O.decode_soa(st)
    sh = plan_interval_shards(soa.ref_id, soa.pos, soa.end_pos, soa.rec_off, len(st), world)[rank]
    p = O.pileup(st[sh["byte_lo"]:sh["byte_hi"]])
    keep = clip_columns(p.ref, p.pos, sh["lo"], sh["hi"])
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), ref=p.ref[keep], pos=p.pos[keep], cov=p.cov[keep], nstart=p.nstart[keep], counts=p.counts[keep])
    hist = torch.from_numpy(depth_histogram(p.cov[keep]))
    allreduce_histogram(hist)
    np.save(os.path.join(out_dir, f"h{rank}.npy"), hist.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_contig_split_by_interval_with_exact_halo(tmp_path, world):
    """SURVEY 8(e), refinement for balance: the contigs of one BAM cut into equal-read-count intervals (cuts fall INSIDE
    contigs), every rank piles up its interval plus the exact halo and keeps its own columns; the ranks' columns
    concatenate to the serial result -- n_starting_here and N-skip reads reaching across cuts included -- and the
    reduced depth histogram is the serial one's."""
    n = 9_000
    port = _free_port()
    mp.spawn(_interval_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    st = host_stream(scaled("cfg3", n))
    p = O.pileup(st)
    parts = [np.load(tmp_path / f"s{r}.npz") for r in range(world)]
    for f in ("ref", "pos", "cov", "nstart", "counts"):
        assert np.array_equal(np.concatenate([q[f] for q in parts]), getattr(p, f)), f
    assert all(len(q["pos"]) > 0 for q in parts)
    h = depth_histogram(p.cov); h[0] = 0
    got = np.load(tmp_path / "h0.npy"); got[0] = 0
    assert np.array_equal(got, h)
