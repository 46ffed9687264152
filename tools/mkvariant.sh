#!/bin/bash
# tools/mkvariant.sh NAME "<extra hipcc -D flags>" [SRCDIR]  -> variants/NAME/libhgs_rast.so
# A/B experiments: run with LD_PRELOAD=variants/NAME/libhgs_rast.so (the torch binding then resolves hgs_* there).
# SRCDIR: another csrc directory (e.g. `git archive HEAD humangaussian_amd/csrc include | tar -x -C /tmp/old` to compare commits).
set -e
N=$1; F=$2; R=$(cd "$(dirname "$0")/.." && pwd); D=$R/variants/$N; mkdir -p $D
C=${3:-$R/humangaussian_amd/csrc}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $F -c $C/api.hip -o $D/api.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $F -c $C/render_bwd.hip -o $D/bwd.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC $D/api.o $D/bwd.o -o $D/libhgs_rast.so
rm -f $D/*.o
