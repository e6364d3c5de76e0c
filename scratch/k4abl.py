import sys, os, subprocess, json
for ab in (8192, 16384, 8192+16384, 8192+16384+4096):
    env = dict(os.environ, BGP_K4_ABLATE=str(ab))
    r = subprocess.run([sys.executable, "bench.py", "--reads", "20000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    d = json.loads(r.stdout)
    print("ablate", ab, "k4 ms", d["kernels"]["k4_tile"]["ms_per_launch"], [l for l in r.stderr.splitlines() if "per-wave" in l][-1:], flush=True)
