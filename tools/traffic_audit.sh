#!/bin/bash
# HBM traffic per launch of every kernel tools/mm_bench.py runs (GPU box): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes, median per kernel and grid size, FETCH_SIZE x 2 (gfx950: MI355X_MICROARCH.md), next to the launch's env count.
# A kernel that moves far more than its algorithmic bytes is spilling or re-reading (that is how the spills of the
# multi-minute kernel were found).   usage: tools/traffic_audit.sh
set -e
root=$PWD; out=$root/gpurun_out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/audit_fetch -- python3 $root/tools/mm_bench.py > /dev/null 2> $out/audit_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/audit_write -- python3 $root/tools/mm_bench.py > /dev/null 2> $out/audit_write.err
cd $root
python3 - <<'PY'
import csv, glob, statistics, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kind, pat in (("FETCH_SIZE", "gpurun_out/audit_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "gpurun_out/audit_write/**/*counter_collection.csv")):
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == kind and "t1d::" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:60], int(r["Grid_Size"]))][kind].append(float(r["Counter_Value"]))
print("%-62s %10s %8s %12s %12s" % ("kernel", "grid", "launches", "read MB", "written MB"))
for (k, g), v in sorted(acc.items()):
    rd = statistics.median(v["FETCH_SIZE"]) * 2 * 1024 / 1e6 if v["FETCH_SIZE"] else float("nan")
    wr = statistics.median(v["WRITE_SIZE"]) * 1024 / 1e6 if v["WRITE_SIZE"] else float("nan")
    print("%-62s %10d %8d %12.1f %12.1f" % (k, g, len(v["FETCH_SIZE"]), rd, wr))
PY
