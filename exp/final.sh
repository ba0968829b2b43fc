set -e
root=$PWD; out=$root/gpurun_out; mkdir -p $out
timeout -k 10 400 python bench.py > $out/r03f_bench.json 2> $out/r03f_bench.err
tail -c 400 $out/r03f_bench.json
timeout -k 10 500 bash tools/profile_bench.sh final > $out/r03f_profile.log 2>&1
timeout -k 10 200 python tools/mm_bench.py > $out/r03f_mm.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03f_mm_stats -- python3 $root/tools/mm_bench.py > $out/r03f_mm_prof.log 2>&1
cd $root
timeout -k 10 200 python tools/launch_floor.py f64 > $out/r03f_launch_floor.log 2>&1
T1D_LIB_PATH=$root/exp/libt1d_ab.so timeout -k 10 200 python exp/sm.py > $out/r03f_small_anatomy.log 2>&1
echo done
