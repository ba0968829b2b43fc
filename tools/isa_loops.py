#!/usr/bin/env python3
"""Static look at one kernel's ISA: instruction mix and, per loop, instruction / scratch / SGPR-spill (v_readlane,
v_writelane) / LDS / wait counts.   usage: isa_loops.py <kernel name regex> [hipcc flags...]   (compiles with -save-temps)"""
import collections, os, re, subprocess, sys, tempfile
pat = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-save-temps"] + sys.argv[2:] +
                      ["-o", os.path.join(tmp, "lib.so"), os.path.join(root, "simglucose_amd", "csrc", "t1d_abi.hip")], cwd=tmp, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "t1d_abi-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M):
    name = m.group(1)
    dem = subprocess.check_output(["c++filt", name]).decode().strip()
    if not re.search(pat, dem):
        continue
    a = m.start(); b = s.index(".Lfunc_end", a)
    body = s[a:b].split("\n")
    isins = lambda l: l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")
    ins = [l for l in body if isins(l)]
    c = collections.Counter(l.split()[0] for l in ins)
    print(dem)
    print("  instructions %d  fp64 fma/mul/add %d  ds_read %d  s_waitcnt %d  scratch %d  readlane/writelane %d" % (
        len(ins), c["v_fma_f64"] + c["v_fmac_f64_e32"] + c["v_mul_f64"] + c["v_add_f64"], sum(v for k, v in c.items() if k.startswith("ds_read")),
        c["s_waitcnt"], sum(v for k, v in c.items() if "scratch" in k), c["v_readlane_b32"] + c["v_writelane_b32"]))
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
    for lab, end in sorted(seen.items(), key=lambda kv: labels[kv[0]]):
        seg = [x for x in body[labels[lab]:end] if isins(x)]
        if len(seg) < 200:
            continue
        cc = collections.Counter(x.split()[0] for x in seg)
        print("  loop %-10s lines %5d-%5d instrs %5d scratch %3d lane-spill %3d ds_read %3d waitcnt %3d" % (
            lab, labels[lab], end, len(seg), sum(v for k, v in cc.items() if "scratch" in k), cc["v_readlane_b32"] + cc["v_writelane_b32"],
            sum(v for k, v in cc.items() if k.startswith("ds_read")), cc["s_waitcnt"]))
