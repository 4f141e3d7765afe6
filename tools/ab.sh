#!/bin/bash
# A/B of two builds on the same box: tools/ab.sh <games> <libA> <libB> ... (interleaved, two rounds)
g=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    MZ_LIB=$PWD/$lib timeout -k 10 120 python3 tools/bench_net.py --games $g --iters 20 $ABX || exit 1
  done
done
