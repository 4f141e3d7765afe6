#!/bin/bash
# Host-side AddressSanitizer build of the C ABI (api.cpp only; device code objects as built by `make`) and of the C++ self-play
# driver, for a run on the GPU box:   tools/asan_host.sh   ->  gpurun_abl/asan/{libmuzero_hip.so,selfplay_asan}
#   gpurun -- 'ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 gpurun_abl/asan/selfplay_asan <ttt.f32> 96 3 PLAYOUT fp32 play'
# (modes: play | reanalyse | justreplay | initial; PLAYOUT | HYBRID; fp32 | bf16).  Both are built with ROCm's clang so that one
# sanitizer runtime serves the executable and the library.
set -e
cd "$(dirname "$0")/../muzero_amd/csrc"
mkdir -p ../../gpurun_abl/asan /tmp/asan
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c api.cpp -o /tmp/asan/api_asan.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address -o ../../gpurun_abl/asan/libmuzero_hip.so tree.o net_fp32.o net_mfma.o net_mfma_f16.o /tmp/asan/api_asan.o
cd ../host
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -I../../include selfplay_main.cpp muzero_host.cpp \
    -o ../../gpurun_abl/asan/selfplay_asan -L../../gpurun_abl/asan -lmuzero_hip -Wl,-rpath,'$ORIGIN'
echo built gpurun_abl/asan
