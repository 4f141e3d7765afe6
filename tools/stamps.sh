#!/bin/bash
# Diagnostic build with in-kernel phase stamps (MZ_STAMPS): gpurun_abl/libmz_STAMPS.so; run with MZ_LIB=<that file>.
set -e
cd "$(dirname "$0")/../muzero_amd/csrc"
mkdir -p ../../gpurun_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMZ_STAMPS -c net_mfma.hip -o /tmp/net_mfma_STAMPS.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl/libmz_STAMPS.so tree.o net_fp32.o /tmp/net_mfma_STAMPS.o api.o
