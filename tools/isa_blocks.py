#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel's ISA (hipcc -S --cuda-device-only output cut to the kernel):
    tools/isa_blocks.py kernel.s [min_instructions]
prints, per block, the number of VALU / SALU / VMEM / LDS / other instructions and the most frequent SALU opcodes."""
import collections, re, sys
path = sys.argv[1]; thresh = int(sys.argv[2]) if len(sys.argv) > 2 else 40
blocks = []; cur = ("entry", [])
for line in open(path):
    s = line.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            blocks.append(cur); cur = (m.group(1), [])
        continue
    op = s.split()[0]
    if re.match(r"^[a-z_0-9]+$", op):
        cur[1].append(op)
blocks.append(cur)
def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "s_sleep", "s_setprio")): return "ctl"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"
tot = collections.Counter()
for name, ops in blocks:
    c = collections.Counter(cls(o) for o in ops); tot.update(c)
    if len(ops) >= thresh:
        sal = collections.Counter(o for o in ops if cls(o) == "salu").most_common(8)
        ctl = collections.Counter(o for o in ops if cls(o) == "ctl").most_common(4)
        print("%-12s n %-5d valu %-4d salu %-4d ctl %-4d vmem %-3d lds %-3d smem %-3d | %s | %s" % (
            name, len(ops), c["valu"], c["salu"], c["ctl"], c["vmem"], c["lds"], c["smem"],
            " ".join("%s:%d" % x for x in sal), " ".join("%s:%d" % x for x in ctl)))
print("total", dict(tot))
