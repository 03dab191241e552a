#!/bin/bash
# A second build of the engine library with extra compile flags, for A/B timing through PTM_ENGINE_LIB (tools/ab_libs.sh):
#   bash tools/build_variant.sh p7 -DPTM_PHILOX_ROUNDS=7      ->  ab/libptm_engine_p7.so   (ab/ is git-ignored, travels with gpurun)
# UNITS="ptm_engine ptm_sweep_dp32" limits the recompiled units (the others are taken from the default build's objects).
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/ptmcmc_amd/csrc
O=$R/ab/obj_$NAME
mkdir -p $O
ALL="ptm_engine ptm_sweep_dp4 ptm_sweep_dp8 ptm_sweep_dp16 ptm_sweep_dp32 ptm_sweep_dp64 ptm_sweep_dp128 ptm_sweep_dp256 ptm_sweep_dp512 ptm_sweep_dp1024"
UNITS=${UNITS:-$ALL}
pids=""
for u in $UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off "$@" -c -o $O/$u.o $C/$u.hip &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
objs=""
for u in $ALL; do
  if [ -f $O/$u.o ]; then objs="$objs $O/$u.o"; else objs="$objs $C/build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libptm_engine_$NAME.so $objs
echo "built ab/libptm_engine_$NAME.so"
