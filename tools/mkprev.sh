#!/bin/bash
# Builds the kernels of a committed revision (default HEAD) as variant "prev", for an A/B against the working tree.
# usage: tools/mkprev.sh [revision]
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
T=$(mktemp -d)
git archive $REV simple-ray-tracer_amd/csrc include | tar -x -C $T
V=variants/prev; mkdir -p $V/csrc $V/host
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -o $V/csrc/libsrt_hip.so $T/simple-ray-tracer_amd/csrc/srt_kernels.hip
cp simple-ray-tracer_amd/host/libsrt_host.so $V/host/
rm -rf $T
echo built $V from $REV
