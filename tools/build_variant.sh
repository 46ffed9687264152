#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>" [alternative render_bwd source]
#   -> variants/NAME/libhgs_rast.so (A/B experiments: run with
#      LD_PRELOAD=variants/NAME/libhgs_rast.so so the torch binding resolves hgs_* there)
set -e
N=$1; F=$2; B=${3:-/root/repo/humangaussian_amd/csrc/render_bwd.hip}; D=/root/repo/variants/$N; mkdir -p $D
C=/root/repo/humangaussian_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $F -c $C/api.hip -o $D/api.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-slp-vectorize -I$C $F -c $B -o $D/bwd.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC $D/api.o $D/bwd.o -o $D/libhgs_rast.so
rm -f $D/*.o
