import sys, os, subprocess, json
for sr in (8, 32, 128, 512):
    env = dict(os.environ, BGP_SEG_RECORDS=str(sr))
    r = subprocess.run([sys.executable, "bench.py", "--reads", "50000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    d = json.loads(r.stdout)
    k = d["kernels"]
    print("seg_records", sr, "step ms", d["ms_per_step"], {n: (k[n]["ms_per_launch"], k[n]["launches_per_step"]) for n in ("k0_guess", "k0_chase", "k1_emit")}, flush=True)
