#!/bin/bash
# Experiment build of the library with a variant of the exact-fp32 kernel:
#   tools/f32lib.sh <name> [extra hipcc flags, e.g. -DMZ_FP32_STAMPS -DMZ_X_NOA]   -> gpurun_abl/libmz_<name>.so
# Run with MZ_LIB=$PWD/gpurun_abl/libmz_<name>.so.  Needs `make` done first (the other objects are reused).
set -e
name=$1; shift
cd "$(dirname "$0")/../muzero_amd/csrc"
mkdir -p ../../gpurun_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c net_fp32.hip -o /tmp/nf_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl/libmz_$name.so tree.o /tmp/nf_$name.o net_mfma.o net_mfma_f16.o api.o
echo built gpurun_abl/libmz_$name.so
