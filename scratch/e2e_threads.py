import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from biod_amd.pileup import Context
ctx = Context(0)
path = "/tmp/e2e_file.bam"
ctx.reset(); ctx.load_bam(path, 0)
for nt in (8, 16, 32, 64, 128, 256):
    ctx.reset()
    t0 = time.perf_counter(); ctx.load_bam(path, nt); t1 = time.perf_counter()
    print(f"threads {nt}: load {t1 - t0:.3f} s", flush=True)
