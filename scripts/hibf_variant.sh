#!/bin/bash
# Builds a libganon_hip variant that differs in compile-time switches of csrc/gn_hibf.hip (-D...) into ganon_amd/csrc/_variants/ (it travels
# to the GPU box; delete it afterwards).  usage: scripts/hibf_variant.sh name "-DGN_PACK_PROF"   ->  GANON_HIP_LIB=$PWD/ganon_amd/csrc/_variants/libganon_hip_name.so
cd "$(dirname "$0")/.."
name=$1; flags=$2
OBJ=ganon_amd/csrc/_obj
mkdir -p ganon_amd/csrc/_variants
lib=ganon_amd/csrc/_variants/libganon_hip_$name.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $flags -c ganon_amd/csrc/gn_hibf.hip -o /tmp/gh_$name.o || exit 1
objs=$(ls $OBJ/*.o | grep -v gn_hibf.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $lib $objs /tmp/gh_$name.o || exit 1
echo $lib
