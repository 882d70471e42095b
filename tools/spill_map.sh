#!/bin/bash
# Where does a kernel spill?  tools/spill_map.sh file.hip mangled-prefix [flags]: lists scratch_* instructions with the nearest label.
src=$1; pre=$2; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --cuda-device-only "$@" -S $src -o /tmp/spill_map.s 2>/dev/null
a=$(grep -n "^$pre" /tmp/spill_map.s | head -1 | cut -d: -f1)
awk -v a=$a 'NR>=a { if ($0 ~ /^\.LBB/) lab=$1; if ($0 ~ /s_barrier/) nb++; if ($0 ~ /scratch_/) print NR-a, lab, "barriers_so_far=" nb, $0; if ($0 ~ /\.end_amdhsa_kernel|^\.Lfunc_end/) exit }' /tmp/spill_map.s
awk -v a=$a 'NR>=a { n++; if ($0 ~ /^\.Lfunc_end/) {print "lines", n; exit} }' /tmp/spill_map.s
