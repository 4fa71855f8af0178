#!/bin/bash
# Builds a whole experiment variant of libsfgs.so with extra -D flags into skyfall-gs_amd/sfgs/_exp/lib_<name>.so
# (git-ignored, travels to the GPU box); compare with tools/ab.sh.   usage: tools/build_variant.sh <name> "<flags>"
set -e
name=$1; defs=${2:-}
cd "$(dirname "$0")/../skyfall-gs_amd/csrc"
mkdir -p ../sfgs/_exp _obj/var_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result"
SRCS="api.cpp raster_fwd.hip raster_bwd.hip ssim.hip knn.hip prepass.hip filter3d.hip densify_stats.hip adam.hip sh_eval.hip compact.hip densify.hip"
for f in $SRCS; do
  extra=""; [ $f = raster_bwd.hip ] && extra="-fno-slp-vectorize"
  # only the rasterizer sources see the experiment flags; the other objects are reused from the main build
  case $f in api.cpp|raster_fwd.hip|raster_bwd.hip) ( /opt/rocm/bin/hipcc $FLAGS $extra $defs -x hip -c $f -o _obj/var_$name/$f.o ) & ;;
    *) cp _obj/$f.o _obj/var_$name/$f.o ;; esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../sfgs/_exp/lib_$name.so _obj/var_$name/*.o
echo built skyfall-gs_amd/sfgs/_exp/lib_$name.so
