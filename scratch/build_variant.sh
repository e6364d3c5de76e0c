#!/bin/bash
# builds libbiodgpu with extra -D flags for ONE unit into scratch/<name>.bin (shipped to the GPU box, git-ignored; BIODGPU_SO=scratch/<name>.bin selects it)
# usage: scratch/build_variant.sh name "-DBGP_G_ABL=1" [unit.hip, default pileup.hip]
set -e
cd /root/repo/biod_amd/csrc
name=$1; flags=$2; unit=${3:-pileup.hip}
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-gpu-rdc $flags -x hip -c $unit -o /tmp/var_$name/$unit.o
objs=$(ls ../_obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/$name.bin $objs /tmp/var_$name/$unit.o -lz -ldl -lpthread
echo built scratch/$name.bin
