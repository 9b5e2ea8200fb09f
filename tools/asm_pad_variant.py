#!/usr/bin/env python3
"""Delta-debugging aid (round 6): rebuild ONE translation unit of the test library from its gfx950 assembly with `s_nop` padding inserted behind
every instruction of a kernel's line range, and link a variant library.  Needs the saved temporaries of
    hipcc -v -save-temps ... -c <file>.hip -o <obj>   (run in gpurun_out/st, log in build.log).
    python tools/asm_pad_variant.py <tag> <kernel substring> <first line> <last line> [nop imm]     (lines relative to the kernel's label; -1 -1: no padding)"""
import os, re, shlex, subprocess, sys
ST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "st")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag, pat, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
imm = int(sys.argv[5]) if len(sys.argv) > 5 else 3
os.chdir(ST)
cmds = [l.strip() for l in open("build.log").read().splitlines() if l.startswith(' "')]
dev_s = next(f for f in os.listdir(".") if f.endswith("gfx950.s"))
orig = dev_s + ".orig"
if not os.path.exists(orig):
    open(orig, "w").write(open(dev_s).read())
L = open(orig).read().splitlines()
i0 = next(i for i, l in enumerate(L) if re.match(r"^(_Z\S+):", l) and pat in l)
i1 = next(i for i in range(i0, len(L)) if L[i].startswith(".Lfunc_end"))
out, n = [], 0
ZERO = os.environ.get("ZERO_INIT")                                      # "v1-69,s5-82": registers cleared at the kernel's entry (uninitialised-read hunt)
VGPRS = os.environ.get("VGPRS")                                       # allocate this many vector registers per lane for the kernel whatever it uses
kd0 = next(i for i, l in enumerate(L) if l.strip().startswith(".amdhsa_kernel") and pat in l)
for i, l in enumerate(L):
    if VGPRS and kd0 < i < kd0 + 80 and ".amdhsa_next_free_vgpr" in l and not any(".end_amdhsa_kernel" in x for x in L[kd0:i]):
        l = "\t\t.amdhsa_next_free_vgpr %s" % VGPRS
    if os.environ.get("SGPRS") and kd0 < i < kd0 + 80 and ".amdhsa_next_free_sgpr" in l and not any(".end_amdhsa_kernel" in x for x in L[kd0:i]):
        l = "\t\t.amdhsa_next_free_sgpr %s" % os.environ["SGPRS"]
    out.append(l)
    if ZERO and i == i0:
        for part in ZERO.split(","):
            kind, rng = part[0], part[1:].split("-")
            for r in range(int(rng[0]), int(rng[1]) + 1):
                out.append("\t%s_mov_b32 %s%d, 0" % ("v" if kind == "v" else "s", kind, r))
        out.append("\ts_mov_b64 vcc, 0")
    k = i - i0
    if i0 < i < i1 and lo <= k <= hi and l.startswith("\t") and not l.strip().startswith((";", ".")) and not re.match(r"\s+(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)", l):
        out.append("\ts_nop %d" % imm); n += 1
EDIT = os.environ.get("EDIT")                                         # a python file run with `K` = the kernel's lines (label first), edited in place
if EDIT:
    j0 = next(i for i, l in enumerate(out) if re.match(r"^(_Z\S+):", l) and pat in l)
    j1 = next(i for i in range(j0, len(out)) if out[i].startswith(".Lfunc_end"))
    K = out[j0:j1]
    exec(open(os.path.join(ROOT, EDIT)).read(), {"K": K, "re": re})
    out[j0:j1] = K
open(dev_s, "w").write("\n".join(out) + "\n")
for c in (cmds[3], cmds[4], cmds[5], cmds[7], cmds[8], cmds[9]):
    subprocess.run(shlex.split(c), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
obj = shlex.split(cmds[9])[shlex.split(cmds[9]).index("-o") + 1]
C = os.path.join(ROOT, "fldr-vfi_amd", "csrc")
others = [os.path.join(C, f) for f in sorted(os.listdir(C)) if f.endswith(".t.o") and f != "prep_kernels.t.o"]
lib = os.path.join(ROOT, "tools", "stamps", "libfldr_%s.so" % tag)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print("%s: %d s_nop %d inserted in lines %d..%d of %s (kernel %d lines)" % (lib, n, imm, lo, hi, L[i0][:50], i1 - i0))
