# per-kernel times of the bare 1024-rung ladder (W = 1) for several engine builds: bash tools/w1_ab.sh "<lib> ..."  (default = the tree's)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/w1ab; rm -rf $O; mkdir -p $O
for L in ${1:-default}; do
  n=$(basename $L .so)
  if [ "$L" = "default" ]; then unset PTM_ENGINE_LIB; else export PTM_ENGINE_LIB=$R/$L; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/tools/w1_probe.py > $O/$n.out 2>$O/$n.err
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1); echo $n; head -3 $f | cut -d, -f1-4
done
