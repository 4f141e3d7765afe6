#!/bin/bash
# Experiment build of the library with only the Go 9x9 one-launch tower instantiated (MZ_FAST_BUILD), fp16 operands:
#   tools/fastlib.sh <name> [extra hipcc flags, e.g. -DMZ_STAMPS -DMZ_X=1]   -> gpurun_abl/libmz_<name>.so
# Run with MZ_LIB=$PWD/gpurun_abl/libmz_<name>.so (tools/ab.sh interleaves several on one box).  Needs `make` done first.
set -e
name=$1; shift
cd "$(dirname "$0")/../muzero_amd/csrc"
mkdir -p ../../gpurun_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMZ_FAST_BUILD -DMZ_F16=1 "$@" -c net_mfma.hip -o /tmp/nm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl/libmz_$name.so tree.o net_fp32.o net_mfma.o /tmp/nm_$name.o api.o
echo built gpurun_abl/libmz_$name.so
