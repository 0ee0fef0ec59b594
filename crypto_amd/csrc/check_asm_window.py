"""Build check for the inline-asm operand loads of qf_gemm.hip (QF_LOAD_AT / QF_WAIT_AT, and the
key-fragment ring of the pinned main loop: QF_RD + its counted lgkmcnt waits).

An `asm volatile("global_load_dwordx4 %0, ..." : "=v"(x))` tells hipcc that x is written when the
statement ends, but the data lands later -- only after the matching asm `s_waitcnt vmcnt(N)`.  In
between, the register allocator must neither read, copy, spill nor re-use those VGPRs: a compiler
`v_mov`/`v_accvgpr_write`/scratch store of them would see stale data, a re-use would be clobbered
by the late write.  check_noscratch.py rules out spills; this script rules out everything else by
scanning the emitted ISA (hipcc -save-temps): inside every kernel whose name contains PREFIX, from
each asm global_load to the next asm `s_waitcnt vmcnt`, NO instruction outside an asm block may
name a register the load will write.  The asm `ds_read_b128` of the fragment ring are checked the
same way against the LDS counter: reads retire in order, an asm (or compiler) `s_waitcnt lgkmcnt(n)`
retires all but the n youngest, and until a read has retired nothing outside an asm block may name
its destination registers.

usage: check_asm_window.py FILE.s PREFIX
"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
SINGLE = re.compile(r"\bv(\d+)\b")


def vregs(text):
    regs = set()
    for a, b in RANGE.findall(text):
        regs.update(range(int(a), int(b) + 1))
    for a in SINGLE.findall(RANGE.sub(" ", text)):
        regs.add(int(a))
    return regs


func, in_asm, pending, loads, kernels, bad = None, False, set(), 0, 0, []
lq, lds_reads = [], 0   # un-retired asm LDS reads, oldest first
for ln, raw in enumerate(open(path, errors="replace"), 1):
    line = raw.split(";", 1)[0].strip() if ";;#" not in raw else raw.strip()
    m = re.match(r"^(_Z\w+):", raw)
    if m:
        func, pending, in_asm, lq = m.group(1), set(), False, []
        if prefix in func:
            kernels += 1
        continue
    if func is None or prefix not in func:
        continue
    if line.startswith(".Lfunc_end"):
        if pending or lq:
            bad.append((func, ln, "kernel ends with asm loads still un-waited"))
        func = None
        continue
    if ";;#ASMSTART" in raw:
        in_asm = True
        continue
    if ";;#ASMEND" in raw:
        in_asm = False
        continue
    if not line or line.startswith(".") or line.endswith(":"):
        continue
    if in_asm:
        if line.startswith("global_load") or line.startswith("buffer_load"):
            dst = line.split(None, 1)[1].split(",")[0]
            if not line.endswith("lds"):          # LDS-DMA has no VGPR destination
                pending |= vregs(dst)
                loads += 1
        elif line.startswith("ds_read"):
            lq.append(vregs(line.split(None, 1)[1].split(",")[0]))
            lds_reads += 1
        elif line.startswith("s_waitcnt"):
            if "vmcnt" in line:
                pending = set()
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                del lq[:max(0, len(lq) - int(m.group(1)))]
        continue
    if line.startswith("s_waitcnt"):
        m = re.search(r"lgkmcnt\((\d+)\)", line)
        if m:
            del lq[:max(0, len(lq) - int(m.group(1)))]
        continue
    for regs in lq:
        hit = vregs(line) & regs
        if hit:
            bad.append((func, ln, "%s   touches v%s of an LDS read that has not retired" % (line, sorted(hit))))
    if pending:
        hit = vregs(line) & pending
        if hit:
            bad.append((func, ln, "%s   touches v%s inside the load window" % (line, sorted(hit))))
if not kernels:
    sys.exit("check_asm_window: no %s kernels found in %s" % (prefix, path))
if bad:
    for f, ln, msg in bad[:40]:
        print("check_asm_window: %s line %d: %s" % (f, ln, msg))
    sys.exit(1)
print("check_asm_window: %d %s kernels, %d asm global loads, %d asm LDS reads, no instruction touches a destination register before its wait" % (kernels, prefix, loads, lds_reads))
