#!/bin/bash
# Register, LDS and occupancy figures of every kernel, from the compiler's own remarks.
# usage: tools/kres.sh ["-Dflags"] [kernel-name-filter]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC $1 \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null simple-ray-tracer_amd/csrc/srt_kernels.hip 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" | sed -e 's/.*remark: //' |
  awk -v f="${2:-k_}" '/Function Name/ {show = index($0, f) > 0} show {printf "%s; ", $0} /LDS Size/ && show {print ""}'
