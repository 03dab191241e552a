#!/bin/bash
# rocprofv3 runs for profiles/: kernel trace + stats, then HBM traffic counters in separate passes (MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; never combine --pmc with trace domains other than kernel-trace).
# usage (on the GPU box): bash tools/profile.sh <round-tag>
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu --no-w1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
