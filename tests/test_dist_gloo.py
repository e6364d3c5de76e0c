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
from biod_amd.shard import allreduce_histogram, depth_histogram, strong_cfg, weak_cfg
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
