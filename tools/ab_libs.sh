#!/bin/bash
# A/B on the GPU box: time bench.py's headline launch with alternative builds of the library (T1D_LIB_PATH) and/or
# option sets.  usage: tools/ab_libs.sh "label|lib.so or -|bench args" ...   -> one line per entry: label kernel_ms value
for spec in "$@"; do
  IFS='|' read -r label lib args <<< "$spec"
  if [ "$lib" = "-" ]; then unset T1D_LIB_PATH; else export T1D_LIB_PATH="$lib"; fi
  out=$(python bench.py --no-cpu-baseline --no-accuracy --steps 600 --warmup 200 $args 2>/dev/null)
  python - "$label" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); print("%-28s kernel_us %8.2f  env-steps/s %.3e  frac %.3f sane %s" % (sys.argv[1], j["roofline"]["kernel_ms"] * 1e3, j["value"], j["roofline"]["frac"], j["sane"]))
except Exception as e:
    print(sys.argv[1], "FAILED", sys.argv[2][:200])
PY
done
