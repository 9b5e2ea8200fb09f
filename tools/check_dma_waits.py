#!/usr/bin/env python3
"""Static check of counted `s_waitcnt vmcnt(N)` waits behind LDS-DMA (global_load_lds) in a compiled kernel (gfx950 `-S` listing).

dec23_synth_kernel's consumer waves prove that their LDS-DMA pieces of tile k + 2 have landed with `s_waitcnt vmcnt(3)` in front of the
tile barrier: correct only if at least three vector-memory instructions (the three frame stores) were issued after the last piece on
every path to that wait (vmcnt counts in order), and free only if it is exactly three.  Nothing in the source enforces what the
compiler emits there — a split store, a spill, another hipcc — so the invariant is checked on the listing (CPU test:
tests/test_host_cpu.py::test_dec23_counted_vmcnt_invariant):

  * forward dataflow over the kernel's basic blocks; state = the set of possible numbers of vector-memory instructions issued since the
    wave's last LDS-DMA piece (saturating), or SAFE once a wait has covered it, or NONE (no piece pending);
  * at every `s_barrier` no piece may be pending uncovered (every state SAFE / NONE);
  * at the counted wait the largest pending count is reported (more than N: the wait also waits for stores — legal, slower);
  * the kernel uses no scratch memory (a spill reload is a vmcnt(0) in the middle of the tile).

    python tools/check_dma_waits.py <listing.s> <kernel name substring> [--expect-counted N]
"""
import re
import sys

SAT = 8
VMEM = re.compile(r"^\s+(global_|buffer_|scratch_|flat_)\w+")
DMA = re.compile(r"^\s+(global_load_lds|buffer_load\w*\s.*\blds\b)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s+(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
WAIT = re.compile(r"^\s+s_waitcnt\b(.*)")


def kernels(lines, pattern):
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):\s", l)
        if m and pattern in m.group(1) and "$local" not in m.group(1):
            e = next(j for j in range(i, len(lines)) if re.match(r"^\s+s_endpgm", lines[j]))
            out.append((m.group(1), i + 1, e + 1))
    return out


def analyse(lines, start, end):
    """-> (violations, counted waits: list of (line number, N, sorted pending counts seen there), barriers checked)."""
    # basic blocks
    blocks, cur, name = {}, [], "entry"
    order = []
    for i in range(start, end):
        m = LABEL.match(lines[i])
        if m:
            blocks[name] = cur
            order.append(name)
            name, cur = m.group(1), []
            continue
        cur.append(i)
    blocks[name] = cur
    order.append(name)
    # fall-through successor of every block (none behind an unconditional branch or the end of the program); a conditional or
    # unconditional branch INSIDE a block hands the state AT THE BRANCH to its target (blocks are only split at labels)
    fall = {}
    for k, name in enumerate(order):
        last = blocks[name][-1] if blocks[name] else None                # (the last instruction decides)
        lm = BRANCH.match(lines[last]) if last is not None else None
        ends = bool(lm and lm.group(1) == "s_branch") or bool(last is not None and re.match(r"^\s+s_endpgm", lines[last]))
        fall[name] = order[k + 1] if (not ends and k + 1 < len(order)) else None

    def step(state, i, report):
        l = lines[i]
        if DMA.match(l):
            return frozenset([0])
        if VMEM.match(l):
            return frozenset(min(c + 1, SAT) if isinstance(c, int) else c for c in state)
        m = WAIT.match(l)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if v:
                n = int(v.group(1))
                if report is not None and n > 0:
                    report.setdefault(i, (n, set()))[1].update(c for c in state if isinstance(c, int))
                return frozenset("S" if (isinstance(c, int) and n <= c) else c for c in state)
            if re.fullmatch(r"\s*\d+\s*", m.group(1) or ""):      # raw immediate form: treat as vmcnt(0) only if the low bits say so
                imm = int(m.group(1))
                if (imm & 0xF) == 0 and ((imm >> 14) & 3) == 0:
                    return frozenset("S" if isinstance(c, int) else c for c in state)
        return state

    def flow(b, st, out):
        """Run block b from state st; out(target, state) for every edge leaving it."""
        dead = False
        for i in blocks[b]:
            m = BRANCH.match(lines[i])
            if m:
                out(m.group(2), st)
                if m.group(1) == "s_branch":
                    dead = True
                    break
                continue
            if re.match(r"^\s+s_endpgm", lines[i]):
                dead = True
                break
            st = step(st, i, None)
        if not dead and fall[b] is not None:
            out(fall[b], st)

    inb = {name: frozenset() for name in order}
    inb["entry"] = frozenset(["N"])
    work = ["entry"]
    while work:
        b = work.pop()

        def out(target, st):
            if target in inb and not st <= inb[target]:
                inb[target] = inb[target] | st
                work.append(target)
        flow(b, inb[b], out)
    violations, counted, barriers = [], {}, 0
    for b in order:
        st = inb[b]
        if not st:
            continue
        for i in blocks[b]:
            if re.match(r"^\s+s_barrier", lines[i]):
                barriers += 1
                bad = sorted(c for c in st if isinstance(c, int))
                if bad:
                    violations.append("line %d: s_barrier reached with an LDS-DMA piece pending behind only %s vector-memory instruction(s) and no covering wait" % (i + 1, bad))
            st = step(st, i, counted)
    waits = [(i + 1, n, sorted(c)) for i, (n, c) in sorted(counted.items())]
    for ln, n, cs in waits:
        if any(c < n for c in cs):
            violations.append("line %d: s_waitcnt vmcnt(%d) with only %s vector-memory instruction(s) issued after the last LDS-DMA piece on some path" % (ln, n, [c for c in cs if c < n]))
    return violations, waits, barriers


def scratch_bytes(lines, kname):
    for i, l in enumerate(lines):
        if l.strip().startswith(".amdhsa_kernel") and kname in l:
            for j in range(i, min(i + 80, len(lines))):
                m = re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", lines[j])
                if m:
                    return int(m.group(1))
    return None


def check(path, pattern, expect_counted=None):
    lines = open(path).read().splitlines()
    ks = kernels(lines, pattern)
    if not ks:
        return ["no kernel matching %r in %s" % (pattern, path)], []
    problems, notes = [], []
    for name, s, e in ks:
        v, waits, barriers = analyse(lines, s, e)
        sc = scratch_bytes(lines, name)
        notes.append("%s: %d barriers checked, counted waits behind LDS-DMA %s, scratch %s B" % (name, barriers, [(n, cs) for _, n, cs in waits if cs], sc))
        problems += ["%s: %s" % (name, x) for x in v]
        if sc:
            problems.append("%s: %d bytes of scratch memory" % (name, sc))
        if expect_counted is not None:
            hits = [(n, cs) for _, n, cs in waits if n == expect_counted and cs]
            if not hits:
                problems.append("%s: no s_waitcnt vmcnt(%d) behind an LDS-DMA piece found (the structure changed: re-derive the count)" % (name, expect_counted))
            for n, cs in hits:
                if max(cs) > n:
                    problems.append("%s: vmcnt(%d) reached with up to %d vector-memory instructions behind the last piece: the wait also waits for %d of them (slower, not wrong)"
                                    % (name, n, max(cs), max(cs) - n))
    return problems, notes


if __name__ == "__main__":
    exp = None
    if "--expect-counted" in sys.argv:
        k = sys.argv.index("--expect-counted")
        exp = int(sys.argv[k + 1])
        del sys.argv[k:k + 2]
    problems, notes = check(sys.argv[1], sys.argv[2], exp)
    print("\n".join(notes))
    if problems:
        print("\n".join("PROBLEM " + p for p in problems))
        sys.exit(1)
    print("OK")
