#!/bin/bash
# A/B library with the 16x16x32 experiment kernel in place of net_mfma.hip: gpurun_abl/libmz_exp16.so (needs `make` done first)
set -e
cd "$(dirname "$0")/../../muzero_amd/csrc"
mkdir -p ../../gpurun_abl
for f in 0 1; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -DMZ_F16=$f "$@" -c ../../tools/experiments/net_mfma_16x16x32.hip -o /tmp/exp16_$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl/libmz_exp16.so tree.o net_fp32.o /tmp/exp16_0.o /tmp/exp16_1.o api.o
echo built gpurun_abl/libmz_exp16.so
