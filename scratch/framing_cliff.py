"""What a framing whose guesses fail costs: cfg3's read model framed with the right reference bound, with a wrong one
(bgp_set_n_ref(1) on a stream of 8 contigs: every header of contigs 1..7 fails the guesser's first pass) -- K0 time."""
import sys, time
sys.path.insert(0, ".")
from biod_amd.pileup import Context
from biod_amd.synth import scaled
ctx = Context(0)
for n in (int(a) for a in sys.argv[1:] or ["4000000"]):
    cfg = scaled("cfg3", n)
    for n_ref in (0, 8, 1):
        ctx.reset()
        ctx.synth_device(cfg)
        if n_ref:
            ctx.set_n_ref(n_ref)
        ctx.timing_enable(True); ctx.timing_reset()
        t0 = time.perf_counter(); ctx.flush(); dt = time.perf_counter() - t0
        t = ctx.timing_get()
        print(f"reads {n} n_ref hint {n_ref}: flush {dt * 1e3:.1f} ms; " + ", ".join(f"{k} {v['total_ms']:.2f} ms x{v['launches']}" for k, v in t.items() if k.startswith('k0')), flush=True)
        assert ctx.num_records() == n
