#!/bin/bash
# Compact per-kernel resource table (VGPR / AGPR / scratch / occupancy) for the csrc/*.hip files given.
# usage: tools/kernel_resources.sh gemm attention ...
cd "$(dirname "$0")/../multi-modal-emotion_amd/csrc" || exit 1
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -c $f.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name/{name=$NF; sub(/\[.*/,"",name); n=$(NF-1)} /remark: +VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /LDS Size/{printf "%-90s vgpr=%-4s agpr=%-4s scratch=%-5s occ=%s\n", n, v, a, s, o}' | while read l; do echo "$l" | sed 's/_ZN3tav//' ; done
done
