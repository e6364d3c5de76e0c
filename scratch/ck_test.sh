#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_chunks or long_reads_checked or reference_base_random or read_dna" 2>&1 | tail -15
BGP_MD_CK_SERIAL=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_chunks" 2>&1 | tail -3
for V in 0 1; do BGP_MD_CK_SERIAL=$V python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['md_path']; print('serial=$V', m['k6_ms'], m['kernels_ms'], m['prefix_vs_oracle']['equal'])"; done
