#!/bin/bash
# Diagnostic builds of the MFMA kernels with one ingredient removed (results are WRONG on purpose;
# only the timing is read).  Usage: tools/ablate.sh  (builds), then run tools/bench_net.py with MZ_LIB.
set -e
cd "$(dirname "$0")/../muzero_amd/csrc"
mkdir -p ../../gpurun_abl
for A in NOMFMA NORING NOBARRIER; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMZ_ABL_$A -c net_mfma.hip -o /tmp/net_mfma_$A.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl/libmz_$A.so tree.o net_fp32.o /tmp/net_mfma_$A.o api.o
done
