#!/bin/bash
# Prints VGPR / scratch / occupancy / LDS of every kernel in the given .hip files (default: all).
cd "$(dirname "$0")/../parquet-go_amd/csrc" || exit 1
files=${@:-bitunpack.hip pages.hip segdelta.hip seghybrid.hip bytearray.hip assemble.hip walk.hip}
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -x hip -c $f -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep remark | sed 's/.*remark: *//; s/ \[-Rpass.*//' |
    awk '/Function Name/{name=$3} /^VGPRs:/{v=$2} /ScratchSize/{s=$3} /Occupancy/{o=$3} /LDS Size/{printf "%-90s vgpr=%s scratch=%s occ=%s lds=%s\n", substr(name,1,90), v, s, o, $4}' | c++filt | cut -c1-40,120-400
done
