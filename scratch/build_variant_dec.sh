#!/bin/bash
# builds libbiodgpu with extra -D flags for decode.hip into scratch/<name>.bin (shipped to the GPU box, git-ignored)
# usage: scratch/build_variant_dec.sh name "-DBGP_SD_PROF"
set -e
cd /root/repo/biod_amd/csrc
name=$1; flags=$2
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $flags -x hip -c decode.hip -o /tmp/var_$name/decode.hip.o
objs=$(ls ../_obj/*.o | grep -v decode.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/$name.bin $objs /tmp/var_$name/decode.hip.o -lz -ldl -lpthread
echo built scratch/$name.bin
