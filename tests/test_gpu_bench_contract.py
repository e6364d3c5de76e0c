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
