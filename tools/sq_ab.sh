#!/bin/bash
# SQ counters of the headline launch under the timing-only batch flags of a tuning build (GPU box).  Needs a -DT1D_AB_FLAGS=1
# build of the library at exp/libt1d_ab.so (hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC
# -DT1D_AB_FLAGS=1 -Iinclude -o exp/libt1d_ab.so simglucose_amd/csrc/t1d_abi.hip; exp/ is git-ignored scratch that travels to the box).
export T1D_LIB_PATH=$PWD/exp/libt1d_ab.so
root=$PWD; out=$root/gpurun_out; cd /tmp; export TMPDIR=/tmp
for cfg in "0 --opt levels=2" "0" "2000"; do
  tag=$(echo $cfg | tr -d ' =-'); 
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/sq3_$tag -- python3 $root/tools/ab_flags.py $cfg --prewarm 0 --steps 20 --warmup 5 > /dev/null 2> $out/sq3_$tag.err || exit 1
done
cd $root
python3 - <<'PY'
import csv, glob, statistics, collections
for d in sorted(glob.glob("gpurun_out/sq3_*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "step1" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(d)
    for k, v in sorted(acc.items()):
        print("   %-42s %-22s %.4g (n=%d)" % (k[0], k[1], statistics.median(v), len(v)))
PY
