#!/bin/bash
# The round's profile set, on the GPU box: tools/profile_round.sh <tag>  -> gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/)
# Every GPU step under its own time limit; the script stops at the first failure.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --fp32-games 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split -- python3 $R/bench.py $B > $OUT/bench_split_under_rocprof.log 2>&1
echo split done
export MZ_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $R/bench.py $B > $OUT/bench_serial_under_rocprof.log 2>&1
unset MZ_STREAMS
echo serial done
cp $(find $OUT/split -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
cp $(find $OUT/serial -name '*kernel_stats.csv' | head -1) $OUT/bench_serial_kernel_stats.csv
rm -rf $OUT/split $OUT/serial
bash $R/tools/pmc.sh $1 16384 go9 f16 > $OUT/pmc.log 2>&1
# the bench lines below quote roofline.traffic from THIS measurement (bench.py:find_traffic takes the newest profiles/*traffic*)
cp $R/gpurun_out/pmc_$1/traffic_go9_16384.json $R/profiles/$1_traffic_go9_16384.json
cp $R/gpurun_out/pmc_$1/traffic_go9_16384.json $OUT/$1_traffic_go9_16384.json
echo pmc done
# the exact-fp32 path (the reference's own arithmetic) as the whole bench: kernel stats for k_net_fp32
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp32 -- python3 $R/bench.py --precision fp32 --fp32-games 0 --no-cpu-baseline > $OUT/bench_fp32_under_rocprof.log 2>&1
cp $(find $OUT/fp32 -name '*kernel_stats.csv' | head -1) $OUT/fp32_kernel_stats.csv
rm -rf $OUT/fp32
echo fp32 done
cd $R
for wl in tictactoe go5 pegsolitair; do
  timeout -k 10 300 python3 bench.py --workload $wl --games 4096 --no-cpu-baseline --fp32-games 0 > $OUT/bench_$wl.log 2>&1
done
timeout -k 10 400 python3 bench.py > $OUT/bench_default.log 2>&1
echo all done
