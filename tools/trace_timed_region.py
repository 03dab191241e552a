#!/usr/bin/env python3
"""From a tools/profile.sh run (gpurun_out/prof_<tag>/): the sweep kernel's launch durations in the rocprofv3 kernel trace -- all of them
(what the stats file averages) and the last `steps` (the bench's timed region) -- beside the figures the bench line of the same process
printed from its own HIP events.  Writes profiles/<tag>_kernel_trace_timed_region.json.
usage: python tools/trace_timed_region.py r04 [steps=20]"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
trace = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
line = [l for l in open(os.path.join(src, "bench_trace.json")).read().splitlines() if l.startswith("{")][-1]
bench = json.loads(line)
kernel = bench["roofline"]["kernel"]
rows = [r for r in csv.DictReader(open(trace)) if kernel in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
summ = lambda d: {"launches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d)}
out = {
    "note": "rocprofv3 --kernel-trace of `bench.py --steps %d --warmup 5 --no-cpu --no-w1`: durations of the sweep kernel's launches from the trace; "
            "`all` = what profiles/%s_kernel_stats.csv averages (settling steps and the launches behind the calibration's idle gap included), "
            "`timed_region` = the last %d launches = the bench's timed steps, to be compared with the line's own HIP-event figures" % (steps, tag, steps),
    "kernel": rows[0]["Kernel_Name"],
    "all": summ(dur),
    "timed_region": summ(dur[-steps:]),
    "bench_line_under_the_profiler": {k: bench["roofline"][k] for k in ("kernel_avg_ms", "kernel_min_ms", "kernel_median_ms", "kernel_max_ms", "frac", "frac_of_measured_copy")},
    "calibration_of_that_run": bench.get("calibration"),
    "value": bench["value"], "ms_per_step": bench["ms_per_step"],
}
dst = os.path.join(ROOT, "profiles", "%s_kernel_trace_timed_region.json" % tag)
json.dump(out, open(dst, "w"), indent=1)
print(dst, json.dumps({k: out[k] for k in ("all", "timed_region", "bench_line_under_the_profiler")}))
