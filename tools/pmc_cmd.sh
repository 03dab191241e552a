#!/bin/bash
# Issue / LDS / memory-instruction counters of the ptm:: kernels of a command, in small groups (separate --pmc passes, no trace
# domain beside them).  usage (GPU box): bash tools/pmc_cmd.sh <tag> python3 tools/kbench_shard.py --walkers 16384 --gpus 8 --reps 3
# The PROGRAM ITSELF must follow the tag -- an interpreter running the workload's script, or the workload's binary: under --pmc the
# profiler's preloaded library has initialised the GPU before the program starts, and any hop through env / bash -c / sh -c /
# taskset / numactl / torchrun / a re-exec'ing launcher is then an exec from a process that holds the GPU, which takes the machine
# down on this pool.  Pin or set the environment BEFORE this script, never behind it; such launchers are refused below.
# Writes gpurun_out/<tag>_pmc_cmd_summary.json (copy into profiles/).
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcc_$TAG
rm -rf $OUT; mkdir -p $OUT
PROG=$1; shift
case "$(basename "$PROG")" in
  env|bash|sh|dash|zsh|taskset|numactl|torchrun|nohup|timeout|time|xargs|sudo|stdbuf|nice)
    echo "pmc_cmd.sh: refusing '$PROG' after 'rocprofv3 --pmc ... --': it would exec the real program from a process that already holds the GPU." >&2
    echo "            Name the program itself (python3 <script> ..., or the binary); set environment and pinning before calling this script." >&2
    exit 2;;
esac
if [ "$(basename "$PROG")" = "python3" ] || [ "$(basename "$PROG")" = "python" ]; then
  case "$1" in -m) if [ "$2" = "torch.distributed.run" ] || [ "$2" = "torch.distributed.launch" ]; then echo "pmc_cmd.sh: refusing a launcher that starts the workload in child processes via exec" >&2; exit 2; fi;; esac
fi
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
