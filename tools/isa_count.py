#!/usr/bin/env python3
"""Instruction-class histogram of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: isa_count.py <file.s> <substring of the kernel's mangled name> [top_n]
The loop structure is not unrolled here: the counts are static instructions, read them next to the source's trip counts."""
import sys, collections
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
c = collections.Counter()
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t[0] in ".;/" or t.split(";")[0].strip().endswith(":"):
        continue
    c[t.split()[0]] += 1
g = collections.Counter()
for op, n in c.items():
    if op.startswith("v_mfma"): g["mfma"] += n
    elif op.startswith("v_pk_"): g["valu_packed"] += n
    elif op.startswith("v_") and "f64" in op: g["valu_f64"] += n
    elif op.startswith("v_"): g["valu_other"] += n
    elif op.startswith("s_"): g["salu"] += n
    elif op.startswith("ds_"): g["lds"] += n
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): g["vmem"] += n
    else: g[op] += n
print(lines[start].split(":")[0], "static instructions:", sum(c.values()))
print(dict(g))
for op, n in c.most_common(top):
    print(f"  {op:28s} {n}")
