#!/bin/bash
# Builds a variant of the two product libraries under variants/<name>/ with extra hipcc flags, for A/B runs
# (select it at run time with SRT_LIB_ROOT=variants/<name>).  usage: tools/mkvar.sh name "-Dflags"
set -e
cd "$(dirname "$0")/.."
V=variants/$1; mkdir -p $V/csrc $V/host
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $2 -o $V/csrc/libsrt_hip.so simple-ray-tracer_amd/csrc/srt_kernels.hip
cp simple-ray-tracer_amd/host/libsrt_host.so $V/host/
echo built $V
