#!/bin/bash
# A/B of several builds on the same box: tools/ab.sh <games> <libA> <libB> ... (interleaved, ${ABR:-2} rounds; ABX = extra
# arguments for tools/bench_net.py, ABI = iterations per run)
g=$1; shift
for round in $(seq 1 ${ABR:-2}); do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    MZ_LIB=$PWD/$lib timeout -k 10 120 python3 tools/bench_net.py --games $g --iters ${ABI:-20} $ABX || exit 1
  done
done
