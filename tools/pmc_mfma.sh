#!/bin/bash
# matrix-pipe / L2 counters of the sweep kernel.  (TA_* / TCP_* counters hang rocprofv3 on this pool: left out.)  usage: bash tools/pmc_mfma.sh <tag>
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcm_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --steps 5 --warmup 2 --no-cpu --no-w1"
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" \
         "TCC_BUSY TCC_REQ TCC_HIT TCC_MISS TCC_EA0_RDREQ" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $ARGS > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "sweep" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s mean/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
