#!/usr/bin/env python3
"""Static guard (round 6): no kernel of a built library may contain a packed-fp32 instruction (v_pk_add / mul / fma_f32) of the operand forms that are
not safe on gfx950 while waves of ANOTHER kernel issue matrix instructions on the same SIMD: a source that is a vector-register pair read through
op_sel = 1 (the low half of the result takes the pair's HIGH register) BEHIND an earlier vector-register source.  Measured stand-alone
(tools/ubench/pk_opsel_probe.hip, profiles/r06_pk_opsel_probe.txt; 8.5e9 lane-results per form): `D, A, B op_sel:[0,1]` (add, mul, fma),
`D, A, B op_sel:[0,1] op_sel_hi:[1,0]`, `D, A, 1.0, C op_sel:[0,0,1]`, `D, S, B, C op_sel:[0,0,1]` returned low half = "the selected operand is 0" in lanes
48-63 in ~1e-6 of their results beside matrix instructions (never alone, never beside vector FMAs); op_sel on the FIRST vector source (`[1,0]`,
`D, S, B, C op_sel:[0,1,0]`; `[1,1]` too, which this check flags all the same), on a scalar source, every op_sel_hi form and v_pk_mov_b32 never did.  level0_prep's tap-window build had one such
instruction and wrote wrong pixels with several frame pairs in flight (profiles/r06_prep_concurrency.txt).

    python tools/check_pk_opsel.py <lib.so | object.o | listing.s> [...]      exit 1 and the offenders if any"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PK = re.compile(r"\b(v_pk_(?:add|mul|fma)_f32)\s+(.*?)\s+op_sel:\[([01,]+)\]")


def offenders_in_text(lines, is_label):
    out, cur = [], "?"
    for l in lines:
        m = is_label(l)
        if m:
            cur = m
            continue
        m = PK.search(l)
        if not m:
            continue
        ops = [x.strip() for x in m.group(2).split(",")]
        sel = m.group(3).split(",")
        seen_vector_source = False
        for i, sv in enumerate(sel):                                 # sources in order: ops[1 + i]
            if 1 + i >= len(ops):
                break
            if ops[1 + i].startswith("v"):
                if sv == "1" and seen_vector_source:
                    out.append((cur, l.split("//")[0].strip()))
                    break
                seen_vector_source = True
    return out


def offenders(path):
    if path.endswith(".s"):
        lab = lambda l: (re.match(r"^(_Z\S+|\w+):", l) or [None, None])[1] if re.match(r"^(_Z\S+|[A-Za-z_]\w*):", l) else None
        return offenders_in_text(open(path).read().splitlines(), lab)
    import kernel_resources as KR
    out = []
    for blob in KR.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout
        lab = lambda l: (re.match(r"^[0-9a-f]+ <(\S+)>:", l) or [None, None])[1] if re.match(r"^[0-9a-f]+ <", l) else None
        out += offenders_in_text(txt.splitlines(), lab)
    return out


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        o = offenders(p)
        kernels = sorted(set(k for k, _ in o))
        print("%s: %d instruction(s) of the class in %d kernel(s)" % (p, len(o), len(kernels)))
        for k in kernels:
            ex = next(i for kk, i in o if kk == k)
            print("    %-100s %3d   e.g. %s" % (k[:100], sum(1 for kk, _ in o if kk == k), ex))
        bad += len(o)
    sys.exit(1 if bad else 0)
