#!/bin/bash
# Compile the library with -Rpass-analysis=kernel-resource-usage and print one line per kernel matching $1 (regex on the
# demangled name): VGPRs, scratch bytes per lane, occupancy.  Extra hipcc flags after the pattern.
pat="${1:-.}"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage "$@" \
    -o /tmp/libt1d_res.so "$(dirname "$0")/../simglucose_amd/csrc/t1d_abi.hip" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name)}
       / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1); print name, "vgpr", v, "scratch", s, "occ", o}
       /error/ {print}' | while read n rest; do echo "$(echo $n | c++filt) $rest"; done | grep -E "$pat"
