#!/bin/bash
# the streamed file path by batch size and device buffers (usage: scratch/stream_pad.sh [n_reads])
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for nb in 2; do for mb in 256 512 1024 1536; do echo "== buffers $nb, batch $mb MiB"; BGP_STREAM_BUFFERS=$nb timeout 600 python scratch/stream_timeline.py ${1:-20000000} $mb 2>&1 | grep "Mreads/s"; done; done
