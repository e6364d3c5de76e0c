"""bench.py's contract with whoever runs it: ONE line on stdout, and that line is the JSON object -- also on the N > 1 code path,
where RCCL prints a version banner to the C stdout that is flushed at exit (after the JSON line, if nothing is done about it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("force_comm", [False, True])
def test_bench_prints_one_json_line(force_comm):
    env = dict(os.environ)
    env.pop("BGP_BENCH_FORCE_COMM", None)
    if force_comm:
        env["BGP_BENCH_FORCE_COMM"] = "1"          # process group + RCCL communicator + bgp_hist_reduce on one GPU
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--parity"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["checks_full_size"]["ok"] is True
    # the oracle's check of the result the timed steps left on the device: on every rank's own shard when there are ranks
    assert d["parity_interior"]["all_equal"] is True and d["parity_interior"]["ranges"] >= 2
    # use_md_tag = true (examples/make_pileup.d:16): timed, and its first columns against the oracle's sweep
    assert d["md_path"]["k6_ms"] > 0 and d["md_path"]["prefix_vs_oracle"]["equal"] is True
    if force_comm:
        assert d["hist_reduce_equals_sum_of_ranks"] is True
        assert d["parity_interior"]["all_equal_on_ranks"] == 1


def _one_json_line(p):
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_one_rank_communicator_uses_one_rccl():
    """bench.py's communicator (ncclCommInitRank) and the product's ncclAllReduce must be the same RCCL: the process holds exactly one
    librccl -- the one torch.distributed mapped -- when the N > 1 code path runs (here with one rank on the one GPU)."""
    env = dict(os.environ); env["BGP_BENCH_FORCE_COMM"] = "1"; env.pop("BGP_BENCH_REDUCE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-md"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    d = _one_json_line(p)
    assert d["hist_reduce_equals_sum_of_ranks"] is True and d["reduce"].startswith("bgp_hist_reduce")
    assert len(d["rccl_libraries_mapped"]) == 1, d["rccl_libraries_mapped"]


def test_two_ranks_rehearsed_on_one_gpu():
    """`bench.py --gpus 2` on a box with one GPU: spawn_ranks -> strong_cfg(rank, 2) -> the real kernels on both ranks' shards of ONE BAM
    (split by contig; pileup.d:497-506: a rank's columns are its own contigs') -> interior parity against the oracle on every
    rank -> the depth histogram summed over the ranks (by the script, over gloo: RCCL refuses two ranks on one device) equals the sum of
    the recomputed ones.  A rehearsal of the N > 1 path, not a scaling measurement: the line says so."""
    env = dict(os.environ); env.pop("BGP_BENCH_FORCE_COMM", None); env.pop("BGP_BENCH_REDUCE", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--parity", "--no-md"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    d = _one_json_line(p)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["reads_per_gpu"] == 1_000_000
    assert d["hist_reduce_equals_sum_of_ranks"] is True
    assert d["parity_interior"]["all_equal"] is True and d["parity_interior"]["all_equal_on_ranks"] == 2
    assert d["checks_full_size"]["ok"] is True and d["checks_full_size"]["ok_on_ranks"] == 2
    import torch
    if torch.cuda.device_count() < 2:
        assert d["reduce"].startswith("REHEARSAL")


def test_eight_ranks_rehearsed_on_one_gpu():
    """`bench.py --gpus 8` at 8 M reads on whatever the box has: strong_cfg(rank, 8) -> eight shards of ONE BAM (a contig a rank), the real
    kernels on every shard, interior parity against the oracle on every rank, the depth histogram summed over the ranks == the sum of the
    recomputed ones.  So that the first run on an 8-GPU node cannot fail on plumbing (pileup.d:497-506, reference.d:96-104).
    On a box with fewer than eight GPUs the line says REHEARSAL: nothing here measures config 4."""
    env = dict(os.environ); env.pop("BGP_BENCH_FORCE_COMM", None); env.pop("BGP_BENCH_REDUCE", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--reads", "8000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--parity", "--no-md", "--no-e2e"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    d = _one_json_line(p)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["reads_per_gpu"] == 1_000_000
    assert d["hist_reduce_equals_sum_of_ranks"] is True
    assert d["parity_interior"]["all_equal"] is True and d["parity_interior"]["all_equal_on_ranks"] == 8
    assert d["checks_full_size"]["ok"] is True and d["checks_full_size"]["ok_on_ranks"] == 8
    import torch
    if torch.cuda.device_count() < 8:
        assert d["reduce"].startswith("REHEARSAL")


def test_config4_file_form_rehearsed():
    """Config 4's FILE form (bench_cfg4_file.py): one indexed BAM, three ranks cut by its .bai (shard.plan_file_shards), every rank's regions
    loaded with the blocks inflated on the device (bgp_load_bam_region_device), piled up, the depth histogram summed over the ranks: reads,
    covered columns, summed coverage and the histogram of the WHOLE file, every read and position counted once."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_cfg4_file.py"), "--gpus", "3", "--reads", "600000", "--reps", "1", "--path", "/tmp/test_cfg4_file.bam"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    d = _one_json_line(p)
    assert d["n_gpus"] == 3 and d["equals_whole_file"] is True and d["reads_counted"] == 600_000 and d["Mreads_per_s"] > 0
