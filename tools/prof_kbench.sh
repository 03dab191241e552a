# usage: bash tools/prof_kbench.sh <tag> [kbench args]   -- rocprofv3 kernel stats of tools/kbench.py
TAG=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_kb_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/kbench.py "$@" > $OUT/kbench.out 2> $OUT/err.log
cat $OUT/kbench.out | tail -1 | cut -c1-300
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-7 $f | head -12
