#!/bin/bash
# Builds experiment variants of libsfgs.so that differ only in raster_bwd.hip's -D flags (see its header) into
# skyfall-gs_amd/sfgs/_exp/ (git-ignored, travels to the GPU box); measure with tools/ab.sh.
# usage: tools/ablate_bwd.sh name1:"-DFLAG ..." name2:"..."
set -e
cd "$(dirname "$0")/../skyfall-gs_amd/csrc"
make -s -j8
mkdir -p ../sfgs/_exp _obj/exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -fno-slp-vectorize"
OTHERS=$(ls _obj/*.o | grep -v raster_bwd)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c raster_bwd.hip -o _obj/exp/bwd_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../sfgs/_exp/lib_$name.so $OTHERS _obj/exp/bwd_$name.o
  echo built ../sfgs/_exp/lib_$name.so
done
