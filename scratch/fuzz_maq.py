"""Differential fuzz of K8 (and K5 under it) on the GPU box: random piles of varied depth against oracle/maq_oracle.py.
usage: python scratch/fuzz_maq.py [rounds] [seed0]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import test_gpu_maq as T
from biod_amd.pileup import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
ctx = Context(0)
t0 = time.time(); bad = 0
for k in range(n):
    rng = np.random.default_rng(seed0 + k)
    try:
        deep = bool(k % 2)
        st = T._random_pile(rng, int(rng.integers(100, 1200)), int(rng.integers(200, 3000)), deep=deep)
        T.check_maq(ctx, st, depcorr=float(rng.choice([0.17, 0.3, 0.05])), eta=float(rng.choice([0.03, 0.01])), min_bq=int(rng.choice([0, 5, 13, 30, 41])),
                    use_md=bool(rng.integers(0, 2)), window=int(rng.integers(40, 3000)))
    except Exception as e:
        bad += 1
        print('seed', seed0 + k, 'FAILED:', repr(e)[:300], flush=True)
print(f"{n} rounds, {bad} failures, {time.time() - t0:.1f} s")
