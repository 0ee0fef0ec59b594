"""Build check for the inline-asm operand loads of qf_gemm.hip (QF_LOAD_AT / QF_WAIT_AT, and the
key-fragment ring of the pinned main loops: QF_RD / QF16_RD + their counted lgkmcnt waits).

An `asm volatile("global_load_dwordx4 %0, ..." : "=v"(x))` tells hipcc that x is written when the
statement ends, but the data lands later -- only after the matching asm `s_waitcnt vmcnt(N)`.  In
between, the register allocator must neither read, copy, spill nor re-use those VGPRs: a compiler
`v_mov`/`v_accvgpr_write`/scratch store of them would see stale data, a re-use would be clobbered
by the late write.  check_noscratch.py rules out spills; this script rules out everything else on
the emitted ISA (hipcc -save-temps).  For every kernel whose name contains PREFIX it builds the
control-flow graph (labels, s_branch / s_cbranch_*) and propagates two pieces of state along every
edge until nothing changes:
  * the VGPRs written by asm global loads that no asm `s_waitcnt vmcnt` has covered yet;
  * the queue of asm `ds_read` destinations that have not retired: LDS reads retire in order, an asm
    or compiler `s_waitcnt lgkmcnt(n)` retires all but the n youngest.
Where paths join, the states are merged conservatively (union of the pending registers; queues are
aligned at their youngest entry and united position by position).  No instruction outside an asm
block may name a register in either state.

usage: check_asm_window.py FILE.s PREFIX
"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
SINGLE = re.compile(r"\bv(\d+)\b")
LABEL = re.compile(r"^(\.LBB\w+):")


def vregs(text):
    regs = set()
    for a, b in RANGE.findall(text):
        regs.update(range(int(a), int(b) + 1))
    for a in SINGLE.findall(RANGE.sub(" ", text)):
        regs.add(int(a))
    return frozenset(regs)


def parse_kernels():
    """yields (name, [(line_no, text, in_asm)])"""
    func, body, in_asm = None, [], False
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            func, body, in_asm = m.group(1), [], False
            continue
        if func is None:
            continue
        if raw.startswith(".Lfunc_end"):
            if prefix in func:
                yield func, body
            func = None
            continue
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        line = raw.split(";", 1)[0].strip()
        if not line or (line.startswith(".") and not LABEL.match(line)):
            continue
        body.append((ln, line, in_asm))


def blocks_of(body):
    """split into basic blocks; returns (blocks, succ) with blocks[i] = list of (ln, text, in_asm)"""
    blocks, label_at, cur = [[]], {}, 0
    for ln, line, in_asm in body:
        m = LABEL.match(line)
        if m:
            if blocks[-1]:
                blocks.append([])
            label_at[m.group(1)] = len(blocks) - 1
            continue
        blocks[-1].append((ln, line, in_asm))
        if line.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append([])
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b[-1][1] if b else ""
        if last.startswith("s_branch"):
            s.append(label_at[last.split()[1]])
        elif last.startswith("s_cbranch"):
            s.append(label_at[last.split()[1]])
            if i + 1 < len(blocks):
                s.append(i + 1)
        elif last.startswith(("s_endpgm", "s_setpc")):
            pass
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    return blocks, succ


def merge(a, b):
    """states are (pending frozenset, queue tuple of frozensets, oldest first)"""
    if a is None:
        return b
    pa, qa = a
    pb, qb = b
    n = max(len(qa), len(qb))
    qa = (frozenset(),) * (n - len(qa)) + qa
    qb = (frozenset(),) * (n - len(qb)) + qb
    return (pa | pb, tuple(x | y for x, y in zip(qa, qb)))


def retire(q, line):
    m = re.search(r"lgkmcnt\((\d+)\)", line)
    if m:
        keep = int(m.group(1))
        q = q[len(q) - keep:] if keep < len(q) else q
        if keep == 0:
            q = ()
    return q


def run_block(block, state, report):
    pending, q = state
    loads = reads = 0
    for ln, line, in_asm in block:
        if in_asm:
            if line.startswith(("global_load", "buffer_load")):
                if not line.endswith("lds"):          # LDS-DMA has no VGPR destination
                    pending = pending | vregs(line.split(None, 1)[1].split(",")[0])
                    loads += 1
            elif line.startswith("ds_read"):
                q = q + (vregs(line.split(None, 1)[1].split(",")[0]),)
                reads += 1
            elif line.startswith("s_waitcnt"):
                if "vmcnt" in line:
                    pending = frozenset()
                q = retire(q, line)
            continue
        if line.startswith("s_waitcnt"):
            q = retire(q, line)
            continue
        if report is not None:
            used = vregs(line)
            for regs in q:
                hit = used & regs
                if hit:
                    report.append((ln, "%s   touches v%s of an LDS read that has not retired" % (line, sorted(hit))))
            hit = used & pending
            if hit:
                report.append((ln, "%s   touches v%s inside the load window" % (line, sorted(hit))))
            if line.startswith("s_endpgm") and (pending or q):
                report.append((ln, "kernel ends with asm loads still un-waited"))
    return (pending, q), loads, reads


kernels = loads = reads = 0
bad = []
for func, body in parse_kernels():
    kernels += 1
    blocks, succ = blocks_of(body)
    state_in = [None] * len(blocks)
    state_in[0] = (frozenset(), ())
    work = [0]
    while work:
        i = work.pop()
        out, _, _ = run_block(blocks[i], state_in[i], None)
        for j in succ[i]:
            m = merge(state_in[j], out)
            if m != state_in[j]:
                state_in[j] = m
                work.append(j)
    for i, b in enumerate(blocks):
        if state_in[i] is None:
            continue                                   # unreachable
        rep = []
        _, l, r = run_block(b, state_in[i], rep)
        loads += l
        reads += r
        bad += [(func, ln, msg) for ln, msg in rep]
if not kernels:
    sys.exit("check_asm_window: no %s kernels found in %s" % (prefix, path))
if bad:
    for f, ln, msg in bad[:40]:
        print("check_asm_window: %s line %d: %s" % (f, ln, msg))
    sys.exit(1)
print("check_asm_window: %d %s kernels, %d asm global loads, %d asm LDS reads, no instruction touches a destination register before its wait" % (kernels, prefix, loads, reads))
