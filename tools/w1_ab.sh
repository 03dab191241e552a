cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/w1ab; rm -rf $O; mkdir -p $O
PTM_ENGINE_LIB=$R/ab/lib_5c0d.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/old -- python3 $R/tools/w1_probe.py > $O/old.out 2>$O/old.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/new -- python3 $R/tools/w1_probe.py > $O/new.out 2>$O/new.err
for n in old new; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); echo $n; head -3 $f | cut -d, -f1-4; done
