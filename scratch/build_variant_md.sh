#!/bin/bash
# builds libbiodgpu with extra -D flags for md.hip into scratch/<name>.bin.  usage: scratch/build_variant_md.sh name "-DX=1"
set -e
cd /root/repo/biod_amd/csrc
name=$1; flags=$2
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $flags -x hip -c md.hip -o /tmp/var_$name/md.hip.o
objs=$(ls ../_obj/*.o | grep -v md.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/$name.bin $objs /tmp/var_$name/md.hip.o -lz -ldl -lpthread
echo built scratch/$name.bin
