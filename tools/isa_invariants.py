"""List the VGPRs a kernel's main loop only READS (loop invariants held in vector registers).
usage: python tools/isa_invariants.py kernel.s <loop header label, e.g. .LBB0_54>"""
import re
import sys
lines = open(sys.argv[1]).read().split("\n")
hdr = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2] + ":"))
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out
written, read = set(), set()
for l in lines[hdr:]:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    parts = l.split(None, 1)
    if len(parts) < 2:
        continue
    op, rest = parts
    ops = [o.strip() for o in rest.split(",")]
    stores = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "v_cmp", "global_atomic"))
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        for o in ops:
            read.update(regs(o))
        continue
    if stores:
        for o in ops:
            read.update(regs(o))
        continue
    written.update(regs(ops[0]))
    for o in ops[1:]:
        read.update(regs(o))
inv = sorted(read - written)
print("%d VGPRs only read inside the loop: %s" % (len(inv), inv))
