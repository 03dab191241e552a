#!/usr/bin/env python3
"""Condense a tools/profile.sh run (gpurun_out/prof_<tag>/) into the tracked summaries under profiles/.
usage: python tools/summarize_profile.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
for leg in ("trace", "fetch", "write"):
    p = os.path.join(src, "bench_%s.json" % leg)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_bench_under_rocprof_%s.json" % (tag, leg)))

pmc = {}
for leg, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(os.path.join(src, "pmc_" + leg, "*", "*_counter_collection.csv"))
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        pmc.setdefault(k, {})[name + "_KB_avg"] = sum(v) / len(v)
        pmc[k]["launches"] = len(v)
out = {"note": "rocprofv3 --pmc, one counter per pass (TCC slots), per-launch averages.  FETCH_SIZE/WRITE_SIZE are in KB "
               "(x1024 B).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide "
               "coalesced streaming read, so hbm_read_bytes = 2*FETCH_SIZE*1024; WRITE_SIZE is exact for streaming stores. "
               "Our accesses are 8 B/lane (uncalibrated width per the guide): treat the read figure as an upper bound.",
       "kernels": {}}
for k, v in pmc.items():
    if "ptm::" not in k:
        continue
    rd = 2 * v.get("FETCH_SIZE_KB_avg", 0) * 1024
    wr = v.get("WRITE_SIZE_KB_avg", 0) * 1024
    out["kernels"][k] = dict(v, hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr)
with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as f:
    json.dump(out, f, indent=1)
print(open(os.path.join(dst, "%s_kernel_stats.csv" % tag)).read())
print(json.dumps(out["kernels"], indent=1))
