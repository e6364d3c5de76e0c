"""What in bench.py's process makes file_e2e's load slower (118 ms) than the stand-alone script's (82 ms)?  The bench's own legs, one at a time.
usage: python scratch/e2e_gap3.py A|B|C|D"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench_e2e
from biod_amd.pileup import Context
from biod_amd.synth import host_stream_np, scaled
from biod_amd import _lib as L
mode = sys.argv[1]
n = 10_000_000
cfg = scaled("cfg3", n)
st = host_stream_np(cfg)
ctx = Context(0)
if mode in ("C", "D"):       # the headline workload first: buffers of 100 M reads
    big = scaled("cfg3", 100_000_000)
    ctx.synth_device(big); ctx.flush(decode=L.BGP_FETCH_ALL); ctx.pileup_run(); ctx.reset()
if mode in ("B", "D"):
    d = bench_e2e.device_e2e(ctx, st, n)
    print(mode, "device_e2e", d["Mreads_per_s"], flush=True)
f = bench_e2e.file_e2e(ctx, st, n, int(cfg.n_ref), path="/tmp/gap3.bam")
print(mode, "file_e2e", f["Mreads_per_s"], f["best_s"], f["runs_s"], flush=True)
