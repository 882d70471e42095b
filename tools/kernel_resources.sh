#!/bin/bash
# Register / LDS / occupancy table of every kernel in one translation unit (compiler remarks, no GPU needed).
#   tools/kernel_resources.sh carma_pack_amd/csrc/carma_pt.hip [filter-regex] [extra hipcc flags...]
src=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /VGPRs:/ && !/AGPRs/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /TotalSGPRs:/ {s=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
       /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); print v, a, s, sc, o, l, name}' |
  while read v a s sc o l name; do echo "vgpr=$v agpr=$a sgpr=$s scratch=$sc occ=$o lds=$l $(echo $name | c++filt | cut -c1-70)"; done | grep -E "$pat"
