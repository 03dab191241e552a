"""Print VGPR / spill / occupancy figures of the kernels in a -Rpass-analysis=kernel-resource-usage log.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> log ; python tools/kernel_resources.py log [name filter]"""
import re
import sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split()[0]
    if flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    print(name[:72], "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "spill", g("VGPRs Spill"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"))
