import sys, os, subprocess, json
for ab in (0, 4, 1, 2):
    env = dict(os.environ, BGP_K4_ABLATE=str(ab))
    out = subprocess.run([sys.executable, "bench.py", "--reads", "20000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
    d = json.loads(out)
    print("ablate", ab, "k4 ms", d["kernels"]["k4_tile"]["ms_per_launch"], "k3", d["kernels"]["k3_keys_scan"]["ms_per_launch"], flush=True)
