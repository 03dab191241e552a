# usage: bash tools/ab_libs.sh "<lib1> <lib2> ..." [kbench args]
LIBS=$1; shift
for L in $LIBS; do
  if [ "$L" = "default" ]; then unset PTM_ENGINE_LIB; else export PTM_ENGINE_LIB=$PWD/$L; fi
  python tools/kbench.py "$@" --tag "$L" 2>&1 | tail -1 | cut -c1-200
done
