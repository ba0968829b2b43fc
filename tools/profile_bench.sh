#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of
# their own (MI355X_MICROARCH.md: counters are collected without any trace domain).  The trace pass times 4 000 launches
# behind 400 warm-up ones: rocprofv3's mean covers all 4 400, so the ~6 % slower first few hundred weigh < 1 %.  Usage: tools/profile_bench.sh TAG [bench args]
set -e
tag=${1:-v10}; shift || true
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r01_stats_$tag -- python3 $root/bench.py --prewarm 0 --warmup 400 --steps 4000 --no-cpu-baseline "$@" > $out/bench_prof_$tag.json 2> $out/bench_prof_$tag.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/r01_fetch_$tag -- python3 $root/bench.py --prewarm 0 --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2> $out/fetch_$tag.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/r01_write_$tag -- python3 $root/bench.py --prewarm 0 --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2> $out/write_$tag.err
cd $root
python3 - "$tag" <<'PY'
import csv, glob, json, statistics, sys
tag = sys.argv[1]
def med(pattern, counter, kern):
    vals = []
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kern in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return statistics.median(vals), len(vals)
bench = json.loads(open("gpurun_out/bench_prof_%s.json" % tag).read().strip().splitlines()[-1])
kern = bench["roofline"]["kernel"].split("<")[0].replace("t1d::", "")
f, nf = med("gpurun_out/r01_fetch_%s/**/*counter_collection.csv" % tag, "FETCH_SIZE", kern)
w, nw = med("gpurun_out/r01_write_%s/**/*counter_collection.csv" % tag, "WRITE_SIZE", kern)
n = bench["config"]["envs_per_gpu"]
res = {"FETCH_SIZE_KiB_median": f, "WRITE_SIZE_KiB_median": w, "launches": [nf, nw], "kernel": bench["roofline"]["kernel"],
       "envs": n, "dtype": bench["dtype"], "n_sub": bench["config"]["n_sub"], "minutes": bench["config"]["minutes_per_launch"],
       "integrator": bench["config"]["integrator"],
       "fetch_bytes_corrected_x2": f * 1024 * 2, "write_bytes": w * 1024,
       "traffic_bytes_per_launch": f * 1024 * 2 + w * 1024, "bytes_per_env_step": (f * 1024 * 2 + w * 1024) / n,
       "note": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over `bench.py --steps 20 --warmup 5 --no-cpu-baseline` (tools/profile_bench.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)"}
json.dump(res, open("gpurun_out/traffic_%s.json" % tag, "w"), indent=1)
print(json.dumps(res))
for fn in glob.glob("gpurun_out/r01_stats_%s/**/*kernel_stats.csv" % tag, recursive=True):
    print(open(fn).read()[:1500])
PY
