#!/bin/bash
# Issue / LDS / memory-instruction counters of the ptm:: kernels of ANY command, in small groups (separate --pmc passes, no trace
# domain beside them).  usage (GPU box): bash tools/pmc_cmd.sh <tag> python3 tools/kbench_shard.py --walkers 16384 --gpus 8 --reps 3
# Writes gpurun_out/<tag>_pmc_cmd_summary.json (copy into profiles/).
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcc_$TAG
rm -rf $OUT; mkdir -p $OUT
PROG=$1; shift
ARGS=""
for a in "$@"; do case "$a" in tools/*|bench.py) ARGS="$ARGS $R/$a";; *) ARGS="$ARGS $a";; esac; done
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
         "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  echo "pass $i: $G"
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- $PROG $ARGS > $OUT/g$i.out 2> $OUT/g$i.err || echo "group $i failed"
done
python3 - $OUT $R/gpurun_out/${TAG}_pmc_cmd_summary.json <<'PY'
import csv, glob, sys, collections, json
out, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ptm::" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --pmc, 2-3 counters per pass, per-launch means", "kernels": {}}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    w = e.get("SQ_WAVES", 0)
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            if c in e: e[c + "_per_wave"] = e[c] / w
    res["kernels"][k] = e
json.dump(res, open(dst, "w"), indent=1)
for k, e in res["kernels"].items():
    print(k[:90]); print("   ", {c: ("%.4g" % v) for c, v in e.items() if c.endswith("_per_wave") or c in ("SQ_WAVES", "launches", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")})
PY
