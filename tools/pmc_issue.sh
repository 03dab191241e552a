#!/bin/bash
# Issue / matrix-pipe counters of the bench's kernels, in SMALL groups (an over-wide --pmc set aborts rocprofv3 on this pool:
# "Request exceeds the capabilities of the hardware"; TA_* / TCP_* counters hang it: left out).  Separate passes, no trace domain
# beside the counters.  Writes profiles/<tag>_pmc_issue_summary.json.   usage (GPU box): bash tools/pmc_issue.sh r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmci_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --steps 5 --warmup 2 --no-cpu --no-w1"
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
         "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
         "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  echo "pass $i: $G"
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $ARGS > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
done
# (copy the summary into profiles/ afterwards: only gpurun_out/ comes back from the GPU box)
python3 - $OUT $R/gpurun_out/${TAG}_pmc_issue_summary.json <<'PY'
import csv, glob, sys, collections, json
out, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ptm::" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --pmc, 2-3 counters per pass, per-launch means over the launches of `bench.py --steps 5 --warmup 2` (settling steps included)", "kernels": {}}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    if "SQ_INSTS_MFMA" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e and e["SQ_INSTS_MFMA"] > 0: e["mfma_busy_cycles_per_inst"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_INSTS_MFMA"]
    if "SQ_INSTS_VALU" in e and "SQ_WAVES" in e and e["SQ_WAVES"] > 0: e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
    if "SQ_INSTS_MFMA" in e and "SQ_WAVES" in e and e["SQ_WAVES"] > 0: e["mfma_insts_per_wave"] = e["SQ_INSTS_MFMA"] / e["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0: e["valu_active_over_busy"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_BUSY_CYCLES"]
    res["kernels"][k] = e
json.dump(res, open(dst, "w"), indent=1)
for k, e in res["kernels"].items():
    if "sweep" in k or "decide" in k or "partition" in k:
        print(k[:80]); print("   ", {c: ("%.4g" % v) for c, v in e.items()})
PY
