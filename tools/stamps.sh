#!/bin/bash
# Diagnostic build with in-kernel phase stamps (MZ_STAMPS): gpurun_abl/libmz_STAMPS.so; run with MZ_LIB=<that file>.
set -e
cd "$(dirname "$0")/../muzero_amd/csrc"
mkdir -p ../../gpurun_abl
# the fp16 variant carries the stamps (run tools/bench_net.py --precision f16); the bf16 object is the product's
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMZ_STAMPS -DMZ_F16=1 -DMZ_FAST_BUILD -c net_mfma.hip -o /tmp/net_mfma_STAMPS.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl/libmz_STAMPS.so tree.o net_fp32.o net_mfma.o /tmp/net_mfma_STAMPS.o api.o
