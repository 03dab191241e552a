# usage: bash tools/ab_lib.sh <lib.so> [kbench args]: the default engine build against another build of it
for L in "" "$1"; do
  if [ -n "$L" ]; then export PTM_ENGINE_LIB=$PWD/$L; else unset PTM_ENGINE_LIB; fi
  python tools/kbench.py "${@:2}" --tag "${L:-default}" 2>&1 | tail -1 | cut -c1-330
  python tools/kbench.py "${@:2}" --tag "${L:-default}" 2>&1 | tail -1 | cut -c1-330
done
