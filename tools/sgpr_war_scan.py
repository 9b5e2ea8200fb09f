#!/usr/bin/env python3
"""Scan a gfx950 listing for a scalar register that a VALU instruction READS and that an SALU instruction (or another VALU writing SGPRs)
OVERWRITES within the next K instructions (write-after-read on an SGPR across the vector / scalar pipes).
    python tools/sgpr_war_scan.py <listing.s> <kernel substring> [K]"""
import re, sys
def regs(tok):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b|\bvcc\b", tok):
        if m.group(0) == "vcc": out.add("vcc")
        elif m.group(3) is not None: out.add(int(m.group(3)))
        else: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
def scan(path, pat, K=6):
    L = open(path).read().splitlines()
    i0 = next(i for i, l in enumerate(L) if re.match(r"^(_Z\S+):", l) and pat in l)
    i1 = next(i for i in range(i0, len(L)) if L[i].startswith(".Lfunc_end"))
    ins = [(i, l.strip()) for i, l in enumerate(L[i0:i1]) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    hits = []
    for n, (i, l) in enumerate(ins):
        op = l.split()[0]
        if not op.startswith("v_"): continue
        ops = l[len(op):].split(",")
        writes_sgpr = op.startswith("v_cmp") or op.startswith("v_readfirstlane") or op.startswith("v_readlane")
        srcs = ops[1:] if ops else []
        if op.startswith("v_cmp") and "_e32" in op: srcs = ops      # e32 compares write vcc implicitly: every listed operand is a source
        rd = set()
        for t in srcs: rd |= regs(t)
        if not rd: continue
        for m in range(1, K + 1):
            if n + m >= len(ins): break
            j, l2 = ins[n + m]
            op2 = l2.split()[0]
            if op2.startswith("s_") and not op2.startswith(("s_nop", "s_waitcnt", "s_cbranch", "s_branch", "s_cmp", "s_barrier", "s_sleep", "s_endpgm", "s_setprio")):
                dst = regs(l2[len(op2):].split(",")[0])
                both = rd & dst
                if both: hits.append((i, l, m, l2, sorted(both, key=str)))
            elif op2.startswith("v_cmp") and "_e64" in op2:
                dst = regs(l2[len(op2):].split(",")[0])
                both = rd & dst
                if both: hits.append((i, l, m, l2, sorted(both, key=str)))
    return hits
if __name__ == "__main__":
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    hits = scan(sys.argv[1], sys.argv[2], K)
    for i, l, m, l2, both in hits:
        print("line %5d  %-60s  +%d: %-50s  %s" % (i, l[:60], m, l2[:50], both))
    print("%d write-after-read pairs within %d instructions" % (len(hits), K))
