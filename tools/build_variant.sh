#!/bin/bash
# Builds an EXPERIMENT variant of libsfgs.so into skyfall-gs_amd/sfgs/_exp/lib_<name>.so (git-ignored; travels to the GPU box;
# compare with tools/ab.sh). The shipped sources carry no experiment knobs: a variant is the sources + patches + -D flags,
# built in a scratch copy of skyfall-gs_amd/csrc.
# usage: tools/build_variant.sh <name> "<-D flags>" [patch ...]        (patches: tools/variants/*.patch, applied with -p1
#        from the repo root, in order; "" for no flags)
#   tools/build_variant.sh nop1 "-DSFGS_BWD_ABLATE=2" tools/variants/bwd_lab_r5.patch
#   FLAG_FILES=ssim.hip tools/build_variant.sh ssim_ilp "-mllvm -amdgpu-sched-strategy=max-ilp"
# The objects of files that are not recompiled are COPIED from the main build: run make on the same sources first.
set -e
name=$1; defs=${2:-}; shift; [ $# -gt 0 ] && shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$ROOT/skyfall-gs_amd/csrc
W=$SRC/_obj/var_$name
rm -rf "$W"; mkdir -p "$W/skyfall-gs_amd/csrc" "$W/include" "$ROOT/skyfall-gs_amd/sfgs/_exp"
cp "$SRC"/*.hip "$SRC"/*.cpp "$SRC"/*.h "$W/skyfall-gs_amd/csrc/"; cp "$ROOT"/include/*.h "$W/include/"
for p in "$@"; do ( cd "$W" && patch -s -p1 < "$ROOT/$p" ); done
cd "$W/skyfall-gs_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result"
SRCS="api.cpp raster_fwd.hip raster_bwd.hip composite_bwd.hip ssim.hip knn.hip prepass.hip filter3d.hip densify_stats.hip adam.hip sh_eval.hip compact.hip densify.hip probe.hip"
mkdir -p o
for f in $SRCS; do
  extra=""; [ $f = raster_bwd.hip -o $f = raster_fwd.hip -o $f = ssim.hip ] && extra="-fno-slp-vectorize"
  [ $f = composite_bwd.hip ] && extra="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
  # only what the patches / flags can touch is recompiled; the other objects come from the main build (make first)
  # FLAG_FILES="a.hip b.hip" (environment): the -D / -mllvm flags go to these files only (default: the rasterizer's three + api.cpp)
  flagged=0; for ff in ${FLAG_FILES:-raster_fwd.hip raster_bwd.hip composite_bwd.hip api.cpp}; do [ $f = $ff ] && flagged=1; done
  fdefs=""; [ $flagged = 1 ] && fdefs="$defs"
  if cmp -s $f "$SRC/$f" && [ -z "$fdefs" ] && \
     cmp -s raster_math.h "$SRC/raster_math.h" && cmp -s sfgs_internal.h "$SRC/sfgs_internal.h" && [ -f "$SRC/_obj/$f.o" ]; then
    cp "$SRC/_obj/$f.o" o/$f.o
  else
    ( /opt/rocm/bin/hipcc $FLAGS $extra $fdefs -x hip -c $f -o o/$f.o 2>&1 | grep -v "hip-link" || true ) &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/skyfall-gs_amd/sfgs/_exp/lib_$name.so" o/*.o
echo built skyfall-gs_amd/sfgs/_exp/lib_$name.so
