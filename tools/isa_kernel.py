#!/usr/bin/env python3
"""Static look at one kernel's ISA from a compile with -S: per loop (by back edge) the instruction, scratch (VGPR spill),
v_readlane/v_writelane (SGPR spill), LDS and wait counts, and the line numbers of every scratch instruction, so that
spills can be told apart by where they sit (around an integration loop: paid per minute; in a rare branch: free).
usage: isa_kernel.py <mangled-name substring> [-o kernel.s]"""
import collections, os, re, subprocess, sys, tempfile
pat = sys.argv[1]
out = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
asm = os.path.join(tmp, "t1d.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only",
                       "-o", asm, os.path.join(root, "simglucose_amd", "csrc", "t1d_abi.hip")], stderr=subprocess.DEVNULL)
s = open(asm).read()
isins = lambda l: l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")
for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
    if pat not in m.group(1):
        continue
    a = m.start(); b = s.index(".Lfunc_end", a)
    body = s[a:b].split("\n")
    if out:
        open(out, "w").write("\n".join(body))
    print(subprocess.check_output(["c++filt", m.group(1)]).decode().strip())
    print("  %d instructions" % len([l for l in body if isins(l)]))
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    seen = {}
    for i, l in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seen[mm.group(1)] = max(seen.get(mm.group(1), 0), i)
    for lb, end in sorted(seen.items(), key=lambda kv: labels[kv[0]]):
        seg = [x for x in body[labels[lb]:end] if isins(x)]
        if len(seg) < 100:
            continue
        cc = collections.Counter(x.split()[0] for x in seg)
        print("  loop %-10s lines %5d-%5d instrs %5d scratch %3d lane-spill %3d ds_read %3d ds_write %3d waitcnt %3d" % (
            lb, labels[lb], end, len(seg), sum(v for k, v in cc.items() if "scratch" in k), cc["v_readlane_b32"] + cc["v_writelane_b32"],
            sum(v for k, v in cc.items() if k.startswith("ds_read")), sum(v for k, v in cc.items() if k.startswith("ds_write")), cc["s_waitcnt"]))
    print("  scratch at lines:", [i for i, l in enumerate(body) if "scratch_" in l])
