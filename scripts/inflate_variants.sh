#!/bin/bash
# Builds libganon_hip variants that differ in compile-time constants of csrc/gn_inflate.hip (-D...) and runs scripts/inflate_probe.py on each
# (GANON_HIP_LIB selects the library).  usage: scripts/inflate_variants.sh "name1:-DGI_X=1 -DGI_Y=2" "name2:..." ; results: gpurun_out/inflate_variants.txt
cd "$(dirname "$0")/.."
OBJ=ganon_amd/csrc/_obj
mkdir -p ganon_amd/csrc/_variants
OUT=gpurun_out/inflate_variants.txt
: > $OUT
for v in "base:" "$@"; do
  name=${v%%:*}; flags=${v#*:}
  lib=ganon_amd/csrc/_variants/libganon_hip_$name.so
  if [ ! -f $lib ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $flags -c ganon_amd/csrc/gn_inflate.hip -o /tmp/gi_$name.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v gn_inflate.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $lib $objs /tmp/gi_$name.o || exit 1
  fi
  if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
    echo "### $name  $flags" >> $OUT
    GANON_HIP_LIB=$PWD/$lib python scripts/inflate_probe.py --reads 1000000 --reps 3 --tile 8 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); b=r['best']
print(r['bytes_equal'], 'step_wall_ms', round(b['ms_step_wall'],1), 'decode', round(b['ms_decode'],1), 'prof', [round(x/1000,1) for x in b['prof_ms'][:7]])" >> $OUT
  fi
done
cat $OUT
