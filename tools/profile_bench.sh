#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run from the repo root): kernel trace + stats, then FETCH_SIZE, WRITE_SIZE
# and the SQ instruction counters in passes of their own (MI355X_MICROARCH.md: counters are collected without any trace
# domain).  The trace pass times 4 000 launches behind 400 warm-up ones.  Writes gpurun_out/<round>_*_TAG and
# gpurun_out/traffic_TAG.json (copy into profiles/<round>/traffic.json for bench.py to pick up).
# usage: tools/profile_bench.sh TAG [bench args]
set -e
tag=${1:-v1}; shift || true
round=${T1D_ROUND:-r03}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --no-accuracy"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${round}_stats_$tag -- python3 $root/bench.py --prewarm 0 --warmup 400 --steps 4000 $common "$@" > $out/bench_prof_$tag.json 2> $out/bench_prof_$tag.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${round}_fetch_$tag -- python3 $root/bench.py --prewarm 0 --steps 20 --warmup 5 $common "$@" > /dev/null 2> $out/fetch_$tag.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${round}_write_$tag -- python3 $root/bench.py --prewarm 0 --steps 20 --warmup 5 $common "$@" > /dev/null 2> $out/write_$tag.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/${round}_sq_$tag -- python3 $root/bench.py --prewarm 0 --steps 20 --warmup 5 $common "$@" > /dev/null 2> $out/sq_$tag.err
cd $root
python3 - "$tag" "$round" <<'PY'
import csv, glob, json, statistics, sys
tag, rnd = sys.argv[1], sys.argv[2]
bench = json.loads(open("gpurun_out/bench_prof_%s.json" % tag).read().strip().splitlines()[-1])
kern = bench["roofline"]["kernel"]
def med(pattern, counter):
    vals = []
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and r["Kernel_Name"] == kern:
                vals.append(float(r["Counter_Value"]))
    return (statistics.median(vals) if vals else None), len(vals)
f, nf = med("gpurun_out/%s_fetch_%s/**/*counter_collection.csv" % (rnd, tag), "FETCH_SIZE")
w, nw = med("gpurun_out/%s_write_%s/**/*counter_collection.csv" % (rnd, tag), "WRITE_SIZE")
sq = {c: med("gpurun_out/%s_sq_%s/**/*counter_collection.csv" % (rnd, tag), c)[0]
      for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")}
n = bench["config"]["envs_per_gpu"]
res = {"FETCH_SIZE_KiB_median": f, "WRITE_SIZE_KiB_median": w, "launches": [nf, nw], "kernel": kern,
       "envs": n, "dtype": bench["dtype"], "n_sub": bench["config"]["n_sub"], "minutes": bench["config"]["minutes_per_launch"],
       "integrator": bench["config"]["integrator"], "sq_counters_per_launch_median": sq,
       "valu_insts_per_launch": sq["SQ_INSTS_VALU"],
       "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes over `bench.py --steps 20 --warmup 5` (tools/profile_bench.sh); "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); SQ_INSTS_VALU counts wave-level instructions"}
if f is not None and w is not None:
    res.update({"fetch_bytes_corrected_x2": f * 1024 * 2, "write_bytes": w * 1024,
                "traffic_bytes_per_launch": f * 1024 * 2 + w * 1024, "bytes_per_env_step": (f * 1024 * 2 + w * 1024) / n})
json.dump(res, open("gpurun_out/traffic_%s.json" % tag, "w"), indent=1)
print(json.dumps(res))
for fn in glob.glob("gpurun_out/%s_stats_%s/**/*kernel_stats.csv" % (rnd, tag), recursive=True):
    print(open(fn).read()[:1500])
PY
