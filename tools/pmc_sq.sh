#!/bin/bash
# SQ counters of the dominant kernel over a short bench.py run (a pass of its own, no trace domain; MI355X_MICROARCH.md).
# usage: tools/pmc_sq.sh TAG [bench args]   -> gpurun_out/pmc_sq_TAG.txt (per-launch medians of the kernel with the most launches)
set -e
tag=${1:-x}; shift || true
root=$PWD; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq_$tag -- python3 $root/bench.py --prewarm 0 --steps 30 --warmup 10 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_sq_$tag.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq2_$tag -- python3 $root/bench.py --prewarm 0 --steps 30 --warmup 10 --no-cpu-baseline "$@" > /dev/null 2>> $out/pmc_sq_$tag.err || true
cd $root
python3 - "$tag" <<'PY' | tee gpurun_out/pmc_sq_$tag.txt
import collections, csv, glob, statistics, sys
tag = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq_%s/**/*counter_collection.csv" % tag, recursive=True) + glob.glob("gpurun_out/pmc_sq2_%s/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not vals:
    print("no counters collected"); sys.exit(0)
kern = max(vals, key=lambda k: max(len(v) for v in vals[k].values()) if "step" in k else 0)
print(kern)
m = {c: statistics.median(v) for c, v in vals[kern].items()}
for c in sorted(m):
    print("  %-22s %14.0f" % (c, m[c]))
if "SQ_WAVES" in m and "SQ_INSTS_VALU" in m:
    print("  per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM rd %.0f wr %.0f" % tuple(m.get(k, 0) / m["SQ_WAVES"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
if "SQ_WAVE_CYCLES" in m:
    wc = m["SQ_WAVE_CYCLES"]
    print("  of wave cycles: active VALU %.2f  wait_any %.2f  wait_inst_any %.2f  active_any %.2f" % (m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc))
PY
